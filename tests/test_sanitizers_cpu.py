"""Host-only code of the library under AddressSanitizer + UBSan and ThreadSanitizer (CPU build with clang++; the full
run is tools/run_sanitizers.py -> profiles/r03_sanitizers.txt).  Here: a short pass per sanitizer so that a regression in
the threaded FASTA parser / .mat writer / copy ring shows up in the CPU suite."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phyloligo_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="no clang++ with sanitizer runtimes")
@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_code_under_sanitizers(tmp_path, san):
    subprocess.run(["make", "-C", CSRC, "san", "SAN=" + san], check=True, capture_output=True)
    exe = os.path.join(CSRC, "build", "san_" + san.replace(",", "_"), "san_host_test")
    from tests.fasta_cases import CASES
    runs = []
    for name in ("crlf", "empty_records", "iupac_u_lowercase", "no_final_newline"):
        p = tmp_path / (name + ".fa")
        p.write_bytes(CASES[name])
        runs.append(["fasta", str(p)])
    runs += [["bigfasta", "12"], ["mat", "300", "700", str(tmp_path / "m.mat")], ["fileread", str(tmp_path / "m.mat")],
             ["pwrite", "1501", str(tmp_path / "c.f32")], ["ring", "20"]]
    for args in runs:
        r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (args, r.stderr[-2000:])

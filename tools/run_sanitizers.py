#!/usr/bin/env python3
"""Builds the host-only code of the library (csrc/po_io.cpp: FASTA parser, .mat writer, file reader, D2H copy ring) under
AddressSanitizer + UBSan and under ThreadSanitizer (clang++, CPU build, `make -C phyloligo_amd/csrc san SAN=...`) and runs
the harness csrc/san/san_host_test on: every hand-built FASTA case of tests/ (results also compared with the oracle's
parser), a 64 MiB multi-segment FASTA, a 10^7-entry .mat, a file read and the copy ring.  Writes the log to stdout
(committed as profiles/r03_sanitizers.txt).  Exit status 0 only if every run is clean."""
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "phyloligo_amd", "csrc")


def fnv(h, b):
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def oracle_line(data):
    from oracle import phyloligo_oracle as po
    import numpy as np
    try:
        titles, seqs = po.parse_fasta(data)
    except ValueError:
        return "rejected (text before the first record)"
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype("<u8")
    h = 1469598103934665603
    for t in titles:
        h = fnv(h, t.encode("latin-1"))
    h = fnv(h, off.tobytes())
    h = fnv(h, b"".join(seqs))
    return "records %d seq_bytes %d hash %016x" % (len(seqs), sum(len(s) for s in seqs), h)


def main():
    from tests.fasta_cases import CASES
    from tests.test_host_cpu import FASTA_CASES          # the CPU suite's list (includes CASES)
    cases = {"host_cpu_%02d" % i: d for i, d in enumerate(FASTA_CASES)}
    cases.update(CASES)
    cases["leading_text"] = b"ACGT\n>a\nAC\n"
    failures = 0
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    with tempfile.TemporaryDirectory() as tmp:
        for san in ("address,undefined", "thread"):
            print("==== make san SAN=%s" % san, flush=True)
            subprocess.run(["make", "-C", CSRC, "san", "SAN=" + san], check=True, stdout=subprocess.DEVNULL)
            exe = os.path.join(CSRC, "build", "san_" + san.replace(",", "_"), "san_host_test")

            def run(*args, expect=None):
                nonlocal failures
                p = subprocess.run([exe, *args], capture_output=True, text=True, env=env)
                out = p.stdout.strip().splitlines()
                bad = p.returncode != 0 or "Sanitizer" in p.stderr or "runtime error" in p.stderr
                if expect is not None and (not out or out[-1] != expect):
                    bad = True
                print("  %-60s rc %d  %s%s" % (" ".join(os.path.basename(a) for a in args), p.returncode, out[-1] if out else "",
                                             "   <-- FAILED" if bad else ""), flush=True)
                if bad:
                    failures += 1
                    print(p.stderr[-4000:])

            for name, data in sorted(cases.items()):
                path = os.path.join(tmp, name + ".fa")
                with open(path, "wb") as fh:
                    fh.write(data)
                if len(data) < 1 << 20:
                    run("fasta", path, expect=oracle_line(data))     # same records as the oracle's parser
            run("bigfasta", "64")                                    # 64 MiB: one segment per host thread, checked in the harness
            run("mat", "2500", "4000", os.path.join(tmp, "big.mat"))  # 10^7 entries through the threaded writer
            run("mat", "3", "5", os.path.join(tmp, "small.mat"))
            run("fileread", os.path.join(tmp, "big.mat"))             # 250 MB through the parallel pread reader
            run("pwrite", "3001", os.path.join(tmp, "c.f32"))         # 36 MB container: whole-row and rectangular blocks, 1-8 writers
            run("ring", "96")                                         # 96 MiB through the two-buffer copy ring
    print("==== %d failing run(s); sanitizer reports: %s" % (failures, "none" if failures == 0 else "see above"))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())

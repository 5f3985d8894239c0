"""the multi-rank container writer against the single-process one, same assembly, same machine.
    python tools/container_writer_bench.py [contigs=30000] [ranks=2]
Single process: `python -m phyloligo_amd ... --large memmap`; multi-rank: the same with `--gpus RANKS` and
PO_CLI_REHEARSAL=1 (all ranks share GPU 0 over gloo - a one-GPU box).  PO_CLI_TIMING=1 makes both print the time of
"distances + container"; the files are compared byte for byte."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phyloligo_amd import synthetic  # noqa: E402


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(64 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR") or None) as tmp:
        seq, off = synthetic.contig_bytes(n, 2000, seed=30001)
        fa = os.path.join(tmp, "a.fa")
        with open(fa, "wb") as fh:
            fh.write(synthetic.fasta_bytes(seq, off))
        env = dict(os.environ, PYTHONPATH=ROOT, PO_CLI_TIMING="1", PO_CLI_REHEARSAL="1")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        res = {}
        for name, extra in (("single", []), ("ranks%d" % ranks, ["--gpus", str(ranks)]), ("single_again", [])):
            out = os.path.join(tmp, name + ".f32")
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "phyloligo_amd", "-i", fa, "-k", "4", "-d", "JSD", "--method", "joblib",
                                "--large", "memmap", "-o", out] + extra, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
            wall = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr[-2000:]
            times = [float(x) for x in re.findall(r"distances \+ container ([0-9.]+) s", r.stderr)]
            res[name] = (max(times), wall, os.path.getsize(out), sha(out))
            print("%-14s distances + container %.3f s (slowest rank), process wall %.2f s, %d bytes" % (name, max(times), wall, res[name][2]), flush=True)
            for ln in r.stderr.splitlines():
                if "phyloligo_amd timing" in ln:
                    print("    " + ln.strip())
            if name != "single":
                assert res[name][3] == res["single"][3], "container bytes differ"
                os.remove(out)
        a, b = res["single"][0], res["ranks%d" % ranks][0]
        print("ratio multi-rank / single-process = %.2f  (%d contigs, %.2f GB container, bytes identical)" % (b / a, n, res["single"][2] / 1e9))


if __name__ == "__main__":
    main()

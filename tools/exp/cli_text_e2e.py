"""one-off: the default command line (text .mat) at 12 000 contigs: wall time of the process"""
import os, subprocess, sys, time
sys.path.insert(0, ".")
import numpy as np
from phyloligo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
fa = "/tmp/t.fa"
seq, off = synthetic.contig_bytes(n, 2000, seed=5)
with open(fa, "wb") as fh:
    for i in range(n):
        fh.write(b">c%07d\n" % i); fh.write(seq[off[i]:off[i + 1]].tobytes()); fh.write(b"\n")
for rep in range(3):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "phyloligo_amd", "-i", fa, "-d", "JSD", "--method", "joblib", "-o", "/tmp/t.mat", "--json-stats", "/tmp/t.json"],
                       capture_output=True, text=True)
    wall = time.time() - t0
    import json
    st = json.load(open("/tmp/t.json"))
    print("run %d: rc %d, process wall %.2f s, %.2f GB of text; seconds %s" % (rep, r.returncode, wall, os.path.getsize("/tmp/t.mat") / 1e9,
          {k: round(v, 3) for k, v in st["seconds"].items()}), flush=True)

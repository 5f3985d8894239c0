"""one-off: where the wall time of the C2 command line goes (97 MB FASTA -> 10 GB float32 container)."""
import json, os, subprocess, sys, time
sys.path.insert(0, ".")
import numpy as np
from phyloligo_amd import synthetic
fa = "/tmp/c2.fa"
seq, off = synthetic.contig_bytes(50000, 2000, seed=50001)
with open(fa, "wb") as fh:
    for i in range(50000):
        fh.write(b">c%07d\n" % i)
        s = seq[off[i]:off[i + 1]]
        for a in range(0, len(s), 80):
            fh.write(s[a:a + 80].tobytes() + b"\n")
for rep in range(3):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-X", "importtime", "-m", "phyloligo_amd", "-i", fa, "-p", "1111", "-d", "JSD", "--method", "joblib", "--large", "memmap",
                        "-o", "/tmp/c2.f32", "--json-stats", "/tmp/c2.json"], capture_output=True, text=True, env=dict(os.environ, PO_CLI_TIMING="1", AMD_LOG_LEVEL="0"))
    wall = time.time() - t0
    st = json.load(open("/tmp/c2.json"))
    imp = [l for l in r.stderr.splitlines() if l.startswith("import time:")]
    top = sorted(((int(l.split("|")[1]), l.split("|")[2].strip()) for l in imp[1:]), reverse=True)[:6]
    print("run %d: process wall %.3f s; json: %s" % (rep, wall, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if not isinstance(v, (dict, list, str))}), flush=True)
    print("   slowest imports (cumulative us):", top, flush=True)
    print("   ", [l for l in r.stderr.splitlines() if "timing" in l])

"""one-off: Kount (sliding windows against the whole-genome profile) on a 1 Gb genome of 5 chromosomes: 2 M windows."""
import os, subprocess, sys, time
import numpy as np
path = "/tmp/genome.fa"
rng = np.random.default_rng(9)
alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
t0 = time.time()
with open(path, "wb") as fh:
    for c in range(5):
        L = 200_000_000 + c
        s = alphabet[rng.integers(0, 4, size=L, dtype=np.uint8)]
        s[10_000_000:10_300_000] = ord("N")                      # a gap: gated windows
        fh.write(b">chr%d test\n" % c)
        full = (L // 60) * 60
        body = np.empty((L // 60, 61), dtype=np.uint8); body[:, :60] = s[:full].reshape(-1, 60); body[:, 60] = 10
        body.tofile(fh)
        if L > full:
            s[full:].tofile(fh); fh.write(b"\n")
print("wrote %.2f GB in %.0f s" % (os.path.getsize(path) / 1e9, time.time() - t0), flush=True)
t0 = time.time()
env = dict(os.environ, PO_CLI_TIMING="1")
r = subprocess.run([sys.executable, "-m", "phyloligo_amd.kount", "-i", path, "-d", "JSD", "-W", "/tmp/kount_out"], capture_output=True, text=True, env=env)
print("kount rc %d in %.1f s" % (r.returncode, time.time() - t0)); print(r.stdout[-600:]); print(r.stderr[-600:])
out = [f for f in os.listdir("/tmp/kount_out")][0]
lines = open(os.path.join("/tmp/kount_out", out)).read().splitlines()
print(out, len(lines), "lines;", lines[0], "|", lines[20_000], "|", lines[-1])
vals = np.array([float(l.split("\t")[3]) for l in lines[::1000]])
print("distance range of a sample:", vals.min(), vals.max())

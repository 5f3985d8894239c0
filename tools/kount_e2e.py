"""Kount mirror end to end on a synthetic 20 Mb genome (one record): prototype + 39 990 windows of 5 kb / step 500."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, '.')
from phyloligo_amd import kount, synthetic
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
rng = np.random.default_rng(3)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L, dtype=np.uint8)]
with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, "g.fa")
    with open(fa, "wb") as fh:
        fh.write(b">chr1 synthetic\n")
        raw = seq.tobytes()
        fh.write(b"\n".join(raw[p:p + 80] for p in range(0, L, 80)) + b"\n")
    opts = type("O", (), {"strand": "both", "n_max_freq_in_windows": 0.4})()
    for it in range(3):
        t = time.perf_counter(); proto = kount.compute_whole_composition(fa, "1111", "both"); t1 = time.perf_counter() - t
        t = time.perf_counter(); rows = kount.sliding_windows_distances(fa, proto, "JSD", "1111", 5000, 500, opts); t2 = time.perf_counter() - t
        print("run %d: prototype %.1f ms, %d windows scanned + compared in %.1f ms (mean JSDx1000 %.3f)"
              % (it, t1 * 1e3, len(rows), t2 * 1e3, float(np.mean([r[3] for r in rows]))), flush=True)

"""one-off: po_write_mat_text on a 12 000 x 12 000 matrix (3.6 GB of text)"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from phyloligo_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
m = np.random.default_rng(0).random((n, n))
np.fill_diagonal(m, 0.0)
for rep in range(3):
    t0 = time.time()
    api.write_mat_text("/tmp/t.mat", m)
    dt = time.time() - t0
    sz = os.path.getsize("/tmp/t.mat")
    print("n = %d: %.2f s, %.2f GB of text, %.2f GB/s, %.1f ns per entry" % (n, dt, sz / 1e9, sz / dt / 1e9, dt / (n * n) * 1e9), flush=True)
os.remove("/tmp/t.mat")

"""one-off: Eucl (exact int8 path) with a float32 matrix at C2 (50 000) and C4 (200 000) size: best kernel time of 3 (for ab.sh)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [50000, 200000]:
    seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
    counts, totals = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
    out = torch.empty((n, n), dtype=torch.float32, device="cuda")
    ms = []
    for _ in range(4):
        _, st = ctx.pairwise(counts, totals, "Eucl", out=out, want_stats=True, dtype="float32")
        ms.append(st["kernel_ms"])
    print("n %6d  Eucl float32  kernel best %7.2f ms (%s)  %.2f TB/s of matrix" % (n, min(ms), " ".join("%.2f" % m for m in ms), n * n * 4 / min(ms) / 1e9), flush=True)
    del out

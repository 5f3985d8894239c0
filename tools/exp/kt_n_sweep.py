import sys, numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
for n in (2048, 4096, 8192, 16384, 32768, 50000):
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
    c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
    best = 1e9
    for _ in range(4):
        _, st = ctx.pairwise(c, t, "KT", out=out, want_stats=True)
        best = min(best, st["kernel_ms"])
    T = (n + 255) // 256
    tiles = T * (T + 1) // 2
    gens = -(-tiles // 256)
    print("N=%6d tiles %6d (%4d generations of 256)  kernel %.3f ms  = %.1f us per generation, %.1f us per tile-slot" % (n, tiles, gens, best, best * 1e3 / gens, best * 1e3 * 256 / tiles), flush=True)
    del out

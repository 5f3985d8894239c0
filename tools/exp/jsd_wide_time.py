"""one-off: JSD on 50 000 fixed-length records of 2 / 8 / 12 / 16 kb: the table kernel's narrow and wide layouts against the general kernel"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
n = 50000
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
for length in (2000, 8000, 12000, 16000):
    seq, off = synthetic.contig_bytes(n, length, seed=length)
    counts, totals = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
    res = []
    for kw in ({}, {"table_path": False}):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, st = ctx.pairwise(counts, totals, "JSD", out=out, want_stats=True, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res.append((best, st["kernel_id"]))
    print("%6d-base records, largest count %3d: default %7.2f ms (id %d)   general kernel %7.2f ms (id %d)" % (length, int(counts.max()), res[0][0] * 1e3, res[0][1], res[1][0] * 1e3, res[1][1]), flush=True)
    del counts, totals

"""one-off: Spearman with a float32 matrix at k = 4, 5, 6 (two digit planes at every width): which float32 kernel from which width on"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
for k, n in ((4, 50000), (5, 50000), (6, 20000)):
    seq, off = synthetic.contig_bytes(n, 2000, seed=77)
    c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1" * k, "both")
    row = []
    for dt, tdt in (("float64", torch.float64), ("float32", torch.float32)):
        out = torch.empty((n, n), dtype=tdt, device="cuda")
        best = 1e9
        for _ in range(3):
            _, st = ctx.pairwise(c, t, "SC", out=out, dtype=dt, want_stats=True)
            best = min(best, st["kernel_ms"])
        row.append("%s kernel %7.3f ms" % (dt, best))
        del out
    print("SC k=%d n=%d  " % (k, n) + "   ".join(row), flush=True)

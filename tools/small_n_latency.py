"""Wall time of one device-resident po_pairwise call at small N (launch- and sync-bound regime), per metric."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
for n in (1000, 2000, 5000, 10000):
    seq, off = synthetic.contig_bytes(n, 2000, seed=1001)
    dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
    out = torch.empty((n, n), dtype=torch.float64, device='cuda')
    line = "N=%5d " % n
    t0 = time.perf_counter()
    for _ in range(20):
        counts, totals = ctx.count_profiles(dseq, doff, "1111", "both")
    torch.cuda.synchronize(); line += "| stage1 %.3f ms " % ((time.perf_counter() - t0) / 20 * 1e3)
    for metric in ("Eucl", "JSD", "BC", "SC", "KT"):
        for _ in range(3):
            ctx.pairwise(counts, totals, metric, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.pairwise(counts, totals, metric, out=out)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
        line += "| %s %.3f (k %.3f p %.3f) " % (metric, wall, st["kernel_ms"], st["prep_ms"])
    print(line, flush=True)

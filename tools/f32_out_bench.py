"""Stage-2 time with float32 output (the --large memmap container type) next to float64, N = 50 000."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
n = 50000
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
c, t = ctx.count_profiles(dseq, doff, "1111", "both")
for metric in ("JSD", "Eucl", "BC", "SC", "KT"):
    for dt, tdt in (("float64", torch.float64), ("float32", torch.float32)):
        out = torch.empty((n, n), dtype=tdt, device="cuda")
        best = 1e9
        for _ in range(3):
            _, st = ctx.pairwise(c, t, metric, out=out, dtype=dt, want_stats=True)
            best = min(best, st["total_ms"])
        print("%-5s %s %.2f ms" % (metric, dt, best), flush=True)
        del out

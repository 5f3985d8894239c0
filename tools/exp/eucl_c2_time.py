"""one-off: Eucl and SC at 50 000 C2 records (counts <= 127, one digit plane), float64 and float32 matrix; best and median of 9 total_ms.
For A/B runs of two libraries on one box (tools/exp/ab.sh): the order of the runs matters by some percent, so alternate them."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
n = 50000
outs = {"float64": torch.empty((n, n), dtype=torch.float64, device="cuda"), "float32": torch.empty((n, n), dtype=torch.float32, device="cuda")}
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
for metric in ("Eucl", "SC"):
    row = []
    for dt, out in outs.items():
        ts = []
        for _ in range(9):
            _, st = ctx.pairwise(c, t, metric, out=out, want_stats=True, dtype=dt)
            ts.append(st["total_ms"])
        row.append("%s best %5.2f median %5.2f ms" % (dt, min(ts), sorted(ts)[4]))
    print("%-4s  %s" % (metric, "   ".join(row)), flush=True)

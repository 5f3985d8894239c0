"""one-off: Eucl / SC / BC with float64 and float32 matrices over record counts (either side of the 8 192-record switch to the
host-side tile classes) and word-space sizes (k = 2 .. 6): total_ms best of 3 and pairs/s, to find cliffs in the round-5 paths."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
for k in (2, 3, 4, 5, 6):
    for n in (1000, 4000, 8191, 8192, 20000, 50000):
        if k >= 6 and n > 20000:
            continue
        seq, off = synthetic.contig_bytes(n, 2000, seed=77)
        c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1" * k, "both")
        row = []
        for metric in ("Eucl", "SC", "BC"):
            for dt, tdt in (("float64", torch.float64), ("float32", torch.float32)):
                out = torch.empty((n, n), dtype=tdt, device="cuda")
                best = 1e9
                for _ in range(3):
                    _, st = ctx.pairwise(c, t, metric, out=out, dtype=dt, want_stats=True)
                    best = min(best, st["total_ms"])
                row.append("%s %s %7.3f" % (metric, dt[-2:], best))
                del out
        print("k=%d n=%6d  " % (k, n) + "  ".join(row) + "   (Eucl f32: %.2e pairs/s)" % (n * (n - 1) / 2 / (float(row[1].split()[-1]) * 1e-3)), flush=True)

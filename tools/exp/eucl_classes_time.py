"""one-off: Eucl and SC at 50 000 records for the three count regimes of the exact int8 path - every count <= 127 (C2/C3), the ragged
assembly (counts to ~10 000: two digit planes in every block), C2 with ONE record of counts up to 2 000 000 (three planes in the tiles
of one block row / column) - with a float64 and a float32 matrix; best of 3 total_ms."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
n = 50000
outs = {"float64": torch.empty((n, n), dtype=torch.float64, device="cuda"), "float32": torch.empty((n, n), dtype=torch.float32, device="cuda")}
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
c1, t1 = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
rseq, roff = synthetic.ragged_assembly(n, seed=2024)
c2, t2 = ctx.count_profiles(torch.from_numpy(rseq).cuda(), torch.from_numpy(roff.astype(np.int64)).cuda(), "1111", "both")
c3 = c1.clone(); t3 = t1.clone()
big = torch.from_numpy(np.random.default_rng(3).integers(0, 2_000_000, size=256).astype(np.int32)).cuda()
c3[7] = big.to(c3.dtype); t3[7] = int(big.sum())
del seq, rseq
for name, (c, t) in (("counts <= 127 (C2)", (c1, t1)), ("ragged assembly", (c2, t2)), ("C2 + one record to 2e6", (c3, t3))):
    for metric in ("Eucl", "SC"):
        row = []
        for dt, out in outs.items():
            best = 1e9
            for _ in range(3):
                _, st = ctx.pairwise(c, t, metric, out=out, want_stats=True, dtype=dt)
                best = min(best, st["total_ms"])
            row.append("%s %6.2f ms" % (dt, best))
        print("%-24s %-4s  %s   (largest count %d)" % (name, metric, "   ".join(row), int(c.max())), flush=True)

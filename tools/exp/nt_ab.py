"""All stage-2 kernels at N = 50 000 (best kernel_ms of 3), for A/B builds."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
n = 50000
ctx = pa.Context(0)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
def prof(pattern, seed):
    seq, off = synthetic.contig_bytes(n, 2000, seed=seed)
    return ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), pattern, "both")
def best(c, t, metric, **kw):
    b = 1e9
    for _ in range(3):
        _, st = ctx.pairwise(c, t, metric, out=out, want_stats=True, **kw)
        b = min(b, st["kernel_ms"])
    return b
c, t = prof("1111", 50001)
res = []
for name, metric, kw in (("JSD table", "JSD", {}), ("JSD general", "JSD", {"table_path": False}), ("Eucl int8", "Eucl", {}), ("Eucl f64", "Eucl", {"table_path": False}),
                         ("SC", "SC", {}), ("BC sad k4", "BC", {"pairdot": False}), ("KT fp4", "KT", {}), ("KT int8", "KT", {"pairdot_i8": True})):
    res.append("%s %.2f" % (name, best(c, t, metric, **kw)))
c, t = prof("11011011", 50005)
res.append("BC thermo C5 %.2f" % best(c, t, "BC"))
res.append("BC sad C5 %.2f" % best(c, t, "BC", pairdot=False))
print(" | ".join(res), flush=True)

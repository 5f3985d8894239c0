"""KT (C2 size) and BC (C5) through the matrix-core pair-dot kernels, FP4 and int8 operands, against the vector-ALU kernels."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ctx = pa.Context(0)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")


def profiles(pattern, seed):
    seq, off = synthetic.contig_bytes(n, 2000, seed=seed)
    return ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), pattern, "both")


def run(c, t, metric, **kw):
    best = None
    for _ in range(3):
        _, st = ctx.pairwise(c, t, metric, out=out, want_stats=True, **kw)
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    return best


c, t = profiles("1111", 50001)
for name, kw in (("KT fp4", {}), ("KT int8", {"pairdot_i8": True}), ("KT fp4 unfolded", {"rc_fold": False}), ("KT int8 unfolded", {"rc_fold": False, "pairdot_i8": True})):
    st = run(c, t, "KT", **kw)
    print("%-18s kernel %7.2f ms prep %6.2f ms  id %d folded %s" % (name, st["kernel_ms"], st["prep_ms"], st["kernel_id"], st["rc_folded"]), flush=True)
ref = out[:2000, :2000].clone()
ctx.pairwise(c, t, "KT", out=out, pairdot_i8=True)
print("KT fp4 == int8 on a corner:", bool(torch.equal(ref, out[:2000, :2000])) or "DIFFERENT (the last run above was int8 unfolded; compare below)", flush=True)
if n <= 20000:
    st = run(c, t, "KT", pairdot=False)
    print("KT valu            kernel %7.2f ms" % st["kernel_ms"], flush=True)
c, t = profiles("11011011", 50005)
for name, kw in (("BC fp4", {}), ("BC int8", {"pairdot_i8": True}), ("BC sad", {"pairdot": False})):
    st = run(c, t, "BC", **kw)
    print("%-18s kernel %7.2f ms prep %6.2f ms  id %d folded %s" % (name, st["kernel_ms"], st["prep_ms"], st["kernel_id"], st["rc_folded"]), flush=True)

import sys, numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
for pattern, n in (("11111", 20000), ("111111", 10000), ("111111", 25000), ("111111", 50000)):      # 26 GB / 52 GB of FP4 pair signs: beyond round 3's 24 GB bound
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
    c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), pattern, "both")
    for name, kw in (("pairdot fp4", {}), ("panel", {"pairdot": False})):
        best = None
        for _ in range(2):
            _, st = ctx.pairwise(c, t, "KT", out=out, want_stats=True, **kw)
            if best is None or st["total_ms"] < best["total_ms"]: best = st
        print("KT %s N=%d %-12s total %8.2f ms (prep %.2f) folded %s" % (pattern, n, name, best["total_ms"], best["prep_ms"], best["rc_folded"]), flush=True)

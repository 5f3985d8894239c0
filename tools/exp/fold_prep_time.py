import sys, time
sys.path.insert(0,'.')
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx=pa.Context(0)
seq,off=synthetic.contig_bytes(50000,2000,seed=50001)
c,t=ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(),"1111","both")
out=torch.empty((50000,50000),dtype=torch.float64,device="cuda")
for m in ("JSD","BC"):
    best=None
    for _ in range(5):
        _,st=ctx.pairwise(c,t,m,out=out,want_stats=True)
        if best is None or st["total_ms"]<best["total_ms"]: best=st
    print(m, "prep %.3f kernel %.3f total %.3f id %d" % (best["prep_ms"],best["kernel_ms"],best["total_ms"],best["kernel_id"]))

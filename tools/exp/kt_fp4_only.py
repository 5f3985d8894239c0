import sys, numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
n = 50000
ctx = pa.Context(0)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
for _ in range(3):
    ctx.pairwise(c, t, "KT", out=out)
torch.cuda.synchronize()

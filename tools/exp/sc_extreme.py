import sys; sys.path.insert(0,'/root/repo')
import numpy as np, phyloligo_amd as pa
ctx=pa.Context(0)
# extreme ranks at D = 16384: strictly increasing / decreasing profiles -> r2 = +-(D-1) at the ends
D=16384
a=np.arange(D,dtype=np.uint32); b=a[::-1].copy(); c=np.roll(a,5); d=np.full(D,3,np.uint32); d[0]=0; d[-1]=9
counts=np.stack([a,b,c,d,a]); totals=counts.sum(1).astype(np.uint64)
g,st=ctx.pairwise(counts,totals,"SC",want_stats=True); r,st0=ctx.pairwise(counts,totals,"SC",want_stats=True,table_path=False)
print(st["kernel_id"],st0["kernel_id"]); print(g); print(np.abs(g-r).max())

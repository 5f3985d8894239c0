import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
n = int(sys.argv[1]); 
ctx = pa.Context(0)
print(ctx.device_name)
t=time.time(); seq, off = synthetic.contig_bytes(n, 2000, seed=50001); print('gen', time.time()-t)
dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
for pat in sys.argv[2].split(','):
    torch.cuda.synchronize(); t=time.time()
    counts, totals = ctx.count_profiles(dseq, doff, pat, 'both'); torch.cuda.synchronize(); print('count', pat, time.time()-t)
    t=time.time(); counts, totals = ctx.count_profiles(dseq, doff, pat, 'both'); torch.cuda.synchronize(); print('count2', pat, (time.time()-t)*1e3, 'ms')
    out = torch.empty((n, n), dtype=torch.float64, device='cuda')
    for metric in sys.argv[3].split(','):
        for it in range(3):
            _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
            pairs = n*(n-1)/2
            print(pat, metric, it, st, 'pairs/s %.3e' % (pairs/(st['total_ms']*1e-3)), 'GB/s out %.1f' % (n*n*8/st['kernel_ms']/1e6))

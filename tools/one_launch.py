"""Launch the stage-2 kernel of one metric a few times on a synthetic assembly (for rocprofv3)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
n = int(sys.argv[1]); metric = sys.argv[2]; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pattern = sys.argv[4] if len(sys.argv) > 4 else "1111"
ragged = len(sys.argv) > 5 and sys.argv[5] == "ragged"
notable = len(sys.argv) > 6 and sys.argv[6] == "notable"
ctx = pa.Context(0)
if ragged:
    rng = np.random.default_rng(7)
    lens = rng.integers(1500, 2500, size=n)
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(off[-1]), dtype=np.uint8)]
else:
    seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
counts, totals = ctx.count_profiles(dseq, doff, pattern, 'both')
out = torch.empty((n, n), dtype=torch.float64, device='cuda')
for it in range(iters):
    _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True, table_path=not notable)
    print(metric, n, pattern, 'kernel_ms %.3f total_ms %.3f pairs/s %.4e' % (st['kernel_ms'], st['total_ms'], n*(n-1)/2/(st['total_ms']*1e-3)))

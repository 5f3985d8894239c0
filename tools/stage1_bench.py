"""Stage-1 kernel time for a few patterns / strands on the C2 assembly (rocprofv3 --kernel-trace --stats around it)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
seq, off = synthetic.contig_bytes(50000, 2000, seed=50001)
dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
for pattern, strand in (("1111", "both"), ("1111", "plus"), ("11011011", "both"), ("1101", "both"), ("111111", "both")):
    for _ in range(3):
        ctx.count_profiles(dseq, doff, pattern, strand)
torch.cuda.synchronize()

"""one-off: stage 1 at C2 (50 000 x 2 kb) for k = 1 .. 8 and on the ragged assembly for k = 6 .. 8"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)


def best(dseq, doff, pattern, strand="both", reps=4):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        c, tt = ctx.count_profiles(dseq, doff, pattern, strand)
        torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
        del c, tt
    return b


seq, off = synthetic.contig_bytes(50000, 2000, seed=50001)
dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
for k in range(1, 9):
    t = best(dseq, doff, "1" * k)
    out_bytes = 50000 * 4 ** k * 4
    print("C2     k=%d  %9.1f us   (count matrix %8.1f MB: %5.2f TB/s of output, %5.2f TB/s of sequence)" % (k, t * 1e6, out_bytes / 1e6, out_bytes / t / 1e12, seq.size / t / 1e12), flush=True)
s3, o3 = synthetic.ragged_assembly(50000, seed=2024)
d3, f3 = torch.from_numpy(s3).cuda(), torch.from_numpy(o3.astype(np.int64)).cuda()
for k in (4, 6, 7, 8):
    t = best(d3, f3, "1" * k)
    out_bytes = 50000 * 4 ** k * 4
    print("ragged k=%d  %9.1f us   (count matrix %8.1f MB: %5.2f TB/s of output, %5.2f TB/s of sequence)" % (k, t * 1e6, out_bytes / 1e6, out_bytes / t / 1e12, s3.size / t / 1e12), flush=True)

"""One launch of every stage-2 tile kernel at N = 50 000 (for rocprofv3 --pmc passes; see tools/pmc_busy.py)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic

ctx = pa.Context(0)
n = 50000
out = torch.empty((n, n), dtype=torch.float64, device="cuda")


def profiles(pattern, seed, ragged=False):      # (ragged=True: lengths 1.5 - 2.5 kb, the round-3 stand-in; unused now)
    if ragged:
        rng = np.random.default_rng(seed)
        lens = rng.integers(1500, 2500, size=n)
        off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(off[-1]), dtype=np.uint8)]
    else:
        seq, off = synthetic.contig_bytes(n, 2000, seed=seed)
    dseq = torch.from_numpy(np.ascontiguousarray(seq)).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
    return ctx.count_profiles(dseq, doff, pattern, "both")


out32 = torch.empty((n, n), dtype=torch.float32, device="cuda")  # the container / multi-GPU CLI type (round 5)
c, t = profiles("1111", 50001)
for metric in ("JSD", "Eucl", "BC", "SC", "KT"):
    ctx.pairwise(c, t, metric, out=out)
ctx.pairwise(c, t, "Eucl", out=out, table_path=False)          # float64 MFMA Gram
for metric in ("Eucl", "SC", "BC"):                             # float32: gram_i8_quad_kernel, gram_i8_half_kernel<SC>, bc_sad_tile_kernel<float>
    ctx.pairwise(c, t, metric, out=out32, dtype="float32")
# the ragged, dirty assembly of bench.py's config.ragged_assembly and tests/test_gpu_full_size.py (every tile a mixed tile)
rseq, roff = synthetic.ragged_assembly(n, seed=2024)
c, t = ctx.count_profiles(torch.from_numpy(rseq).cuda(), torch.from_numpy(roff.astype(np.int64)).cuda(), "1111", "both")
del rseq
ctx.pairwise(c, t, "JSD", out=out)                                # general JSD kernel
ctx.pairwise(c, t, "BC", out=out)                                 # general BC kernel
ctx.pairwise(c, t, "Eucl", out=out)                               # two digit planes
ctx.pairwise(c, t, "Eucl", out=out32, dtype="float32")            # two digit planes, float32: gram_i8_half_kernel<Eucl>
c, t = profiles("11011011", 50005)
ctx.pairwise(c, t, "BC", out=out)                                 # C5: thermometer planes on the matrix cores
ctx.pairwise(c, t, "BC", out=out, pairdot=False)                  # C5 through the packed-byte SAD kernel
torch.cuda.synchronize()
print("done")

"""Stage-1 wall time (3 calls, best) on the C2 assembly for a few patterns / strands, and on a ragged assembly."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
seq, off = synthetic.contig_bytes(50000, 2000, seed=50001)
dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()


def best(dseq, doff, pattern, strand, reps=5):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        c, tt = ctx.count_profiles(dseq, doff, pattern, strand)
        torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    return b, c


which = sys.argv[1] if len(sys.argv) > 1 else "all"
configs = (("1111", "both"), ("1111", "plus"), ("11011011", "both"), ("1101", "both"), ("111111", "both"), ("111111", "minus"),
           ("1111", "minus"), ("1101", "minus"), ("110100111", "both"), ("1011", "both"), ("10011", "both"), ("11101", "both"),
           ("11111", "both"))
if which not in ("all", "ragged", "ragged_dirty"):
    configs = tuple(c for c in configs if "%s_%s" % c == which)
if which in ("ragged", "ragged_dirty"):
    configs = ()
for pattern, strand in configs:
    t, c = best(dseq, doff, pattern, strand)
    nbytes = seq.size + c.numel() * 4 + 50000 * 8
    print("C2 %-9s %-5s %7.1f us  %5.2f TB/s algorithmic (%.0f MB)" % (pattern, strand, t * 1e6, nbytes / t / 1e12, nbytes / 1e6), flush=True)
if which not in ("all", "ragged", "ragged_dirty"):
    sys.exit(0)
if which in ("all", "ragged_dirty"):      # the assembly of bench.py's config.ragged_assembly: + N runs, lower case, IUPAC codes, 4 compositions
    s3, o3 = synthetic.ragged_assembly(50000, seed=2024)
    t, c = best(torch.from_numpy(s3).cuda(), torch.from_numpy(o3.astype(np.int64)).cuda(), "1111", "both")
    nbytes = int(o3[-1]) + c.numel() * 4 + 50000 * 8
    print("ragged_dirty %.2f Gb 1111 both %7.1f us  %5.2f TB/s algorithmic" % (s3.size / 1e9, t * 1e6, nbytes / t / 1e12), flush=True)
    del s3
    if which == "ragged_dirty":
        sys.exit(0)
rng = np.random.default_rng(2024)
n = 50000
lens = np.clip(np.exp(rng.normal(np.log(4000), 1.0, size=n)), 1000, 200000).astype(np.int64)
o2 = np.zeros(n + 1, dtype=np.int64); o2[1:] = np.cumsum(lens)
s2 = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (int(o2[-1]),), device="cuda")]
t, c = best(s2, torch.from_numpy(o2).cuda(), "1111", "both")
nbytes = int(o2[-1]) + c.numel() * 4 + n * 8
print("ragged 0.33 Gb 1111 both %7.1f us  %5.2f TB/s algorithmic" % (t * 1e6, nbytes / t / 1e12), flush=True)

"""one-off: stage 1 on millions of tiny records, either side of the one-launch scan's limit (1 024 workgroups = 1 048 576 records):
counts of sampled records against the oracle, totals against the closed form, both scan paths against each other."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from oracle import phyloligo_oracle as oracle
ctx = pa.Context(0)
rng = np.random.default_rng(12)
for n in (1_048_576, 1_048_577, 4_194_305, 6_000_000):
    lens = rng.integers(0, 41, size=n)                     # records of 0 .. 40 bases
    off = np.zeros(n + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(off[-1]))].copy()
    seq[rng.integers(0, seq.size, size=seq.size // 500)] = ord("N")
    dseq, doff = torch.from_numpy(seq).cuda(), torch.from_numpy(off).cuda()
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c, t = ctx.count_profiles(dseq, doff, "11", "both")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sample = np.concatenate([np.arange(0, 2000), rng.integers(0, n, size=3000), np.arange(n - 2000, n)])
    recs = [seq[off[i]:off[i + 1]].tobytes() for i in sample]
    oc, ot = oracle.compute_counts(recs, "11", "both")
    got_c = c[torch.from_numpy(sample).cuda()].cpu().numpy().astype(np.int64)
    got_t = t[torch.from_numpy(sample).cuda()].cpu().numpy().astype(np.int64)
    ok = np.array_equal(got_c, oc) and np.array_equal(got_t, ot)
    rowsum_ok = bool((c.sum(1, dtype=torch.int64) == t).all())
    print("n = %d (%s scan): %.2f ms, sampled records equal the oracle: %s, every row sums to its total: %s"
          % (n, "one-launch" if (n + 1023) // 1024 <= 1024 else "three-launch", dt * 1e3, ok, rowsum_ok), flush=True)
    del c, t, dseq, doff

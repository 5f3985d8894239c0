"""A real-assembly-like input: 50 000 contigs with log-normal lengths (median 4 kb, 1 kb .. 200 kb, ~0.5 Gb in all),
k = 4, both strands: ragged totals, counts far above 127 - the general JSD / BC kernels, the two-plane int8 Gram."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
ctx = pa.Context(0)
rng = np.random.default_rng(2024)
n = 50000
lens = np.clip(np.exp(rng.normal(np.log(4000), 1.0, size=n)), 1000, 200000).astype(np.int64)
off = np.zeros(n + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
print("contigs %d, bases %.2f Gb, longest %d" % (n, off[-1] / 1e9, lens.max()), flush=True)
seq = torch.randint(0, 4, (int(off[-1]),), dtype=torch.uint8, device="cuda")
seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[seq.long()]
doff = torch.from_numpy(off).cuda()
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    counts, totals = ctx.count_profiles(seq, doff, "1111", "both")
    torch.cuda.synchronize(); s1 = (time.perf_counter() - t) * 1e3
print("stage 1: %.2f ms (%.2f TB/s of sequence bytes), largest count %d" % (s1, off[-1] / s1 / 1e9, int(counts.max())), flush=True)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
pairs = n * (n - 1) / 2
for metric in ("JSD", "Eucl", "BC", "SC", "KT"):
    best = None
    for _ in range(3):
        _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
        if best is None or st["total_ms"] < best["total_ms"]: best = st
    print("%-4s %.2f ms (prep %.2f, kernel id %d, folded %s) %.3e pairs/s" % (metric, best["total_ms"], best["prep_ms"], best["kernel_id"], best["rc_folded"], pairs / (best["total_ms"] * 1e-3)), flush=True)

"""Full matrix at a size beyond the BASELINE configs (default N = 120 000: 115 GB of float64 on one GPU),
spot-checked against the oracle on a few rows and for symmetry.  usage: large_n_check.py [N] [metric] [pattern]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
from oracle import phyloligo_oracle as oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
metric = sys.argv[2] if len(sys.argv) > 2 else "JSD"
pattern = sys.argv[3] if len(sys.argv) > 3 else "1111"
ctx = pa.Context(0)
seq, off = synthetic.contig_bytes(n, 2000, seed=7)
dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
counts, totals = ctx.count_profiles(dseq, doff, pattern, "both")
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter()
_, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
torch.cuda.synchronize()
print("N=%d %s: %.1f ms (kernel %.1f), %.3e pairs/s, kernel id %d, folded %s" % (
    n, metric, st["total_ms"], st["kernel_ms"], n * (n - 1) / 2 / (st["total_ms"] * 1e-3), st["kernel_id"], st["rc_folded"]), flush=True)
freq = oracle.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
rows = [0, n // 3 + 17, n - 1]
cols = np.arange(n) if metric != "KT" else np.sort(np.random.default_rng(1).choice(n, size=64, replace=False))   # the KT oracle is O(D^2) Python per pair
for r in rows:
    w = oracle.pairwise_block(np.vstack([freq[r:r + 1], freq[cols]]), metric, 0, 1)[0, 1:]
    g = out[r].cpu().numpy()[cols]
    r = int(np.searchsorted(cols, r)) if r in cols else -1
    if r >= 0:
        w[r] = 0.0 if metric != "KT" else g[r]
    err = np.nanmax(np.abs(g - w) / np.maximum(np.abs(w), 1e-300) * (np.abs(w) > 1e-12))
    print("row %d: max rel err %.2e" % (r, err), flush=True)
    assert err < 1e-6
idx = torch.randint(0, n, (2, 4096), device="cuda")
assert torch.equal(out[idx[0], idx[1]], out[idx[1], idx[0]])
print("symmetric on 4096 random entries; ok")

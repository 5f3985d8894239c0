"""Time the BASELINE.json configs on one GPU (stage 1 + stage 2), one line per config."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
print(ctx.device_name)
configs = [("C1", 1000, "1111", "Eucl", 1001), ("C2", 50000, "1111", "JSD", 50001), ("C3", 50000, "1111", "Eucl", 50001),
           ("C5", 50000, "11011011", "BC", 50005), ("C2-KT", 50000, "1111", "KT", 50001), ("C2-SC", 50000, "1111", "SC", 50001)]
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
for name, n, pat, metric, seed in configs:
    if only and name not in only: continue
    seq, off = synthetic.contig_bytes(n, 2000, seed=seed)
    dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        counts, totals = ctx.count_profiles(dseq, doff, pat, 'both'); torch.cuda.synchronize()
        s1 = (time.perf_counter() - t) * 1e3
    out = torch.empty((n, n), dtype=torch.float64, device='cuda')
    best = None
    for _ in range(3):
        _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
        if best is None or st['total_ms'] < best['total_ms']: best = st
    pairs = n * (n - 1) / 2
    gen = ""
    if metric in ("JSD", "Eucl", "BC"):
        _, st2 = ctx.pairwise(counts, totals, metric, out=out, want_stats=True, table_path=False)
        _, st2 = ctx.pairwise(counts, totals, metric, out=out, want_stats=True, table_path=False)
        gen = " | general kernel only: %.2f ms %.3e pairs/s" % (st2['total_ms'], pairs / (st2['total_ms'] * 1e-3))
    print("%-6s N=%d pattern=%s D=%d %s: stage1 %.3f ms | prep %.3f ms kernel %.3f ms total %.3f ms | %.3e pairs/s | out %.0f GB/s (kernel id %d)%s"
          % (name, n, pat, counts.shape[1], metric, s1, best['prep_ms'], best['kernel_ms'], best['total_ms'], pairs / (best['total_ms'] * 1e-3),
             n * n * 8 / best['kernel_ms'] / 1e6, best['kernel_id'], gen), flush=True)
    del out, counts, totals, dseq
    torch.cuda.empty_cache()

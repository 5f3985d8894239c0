"""BASELINE config 4 (200 000 contigs, JSD, 8 row blocks) rehearsed on ONE GPU: every rank's tournament work list is
run in turn through po_pairwise_blocks_dev with its real slab (25 000 x 200 000 float64 = 40 GB) and mirror
buffers; entries are spot-checked against the oracle, each unordered pair must be produced exactly once, and the
per-rank times give the load balance.  usage: c4_virtual_ranks.py [N] [world] [metric]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
from phyloligo_amd.dist import RowBlockPlan
from oracle import phyloligo_oracle as oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
metric = sys.argv[3] if len(sys.argv) > 3 else "JSD"
ctx = pa.Context(0)
seq, off = synthetic.contig_bytes(n, 2000, seed=200001)
dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
counts, totals = ctx.count_profiles(dseq, doff, "1111", "both")
del dseq
freq = oracle.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
plan = RowBlockPlan(n, world)
print(plan.describe(), flush=True)
rng = np.random.default_rng(0)
total_pairs = 0
times = []
for g in range(world):
    slab, mirrors = plan.allocate(g, counts.device, torch.float64)
    best = None
    for rep in range(2):                    # the second pass runs with workspaces and page tables warm
        torch.cuda.synchronize(); t = time.perf_counter()
        st = plan.compute(ctx, counts, totals, metric, g, slab, mirrors, want_stats=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
        best = dt if best is None else min(best, dt)
    dt = best
    times.append(dt)
    total_pairs += plan.pair_evaluations(g)
    lo, hi = plan.rows(g)
    worst = 0.0
    for ((r0, r1), (c0, c1), kind, peer), m in zip(plan.work(g), mirrors):
        for _ in range(6):
            i = int(rng.integers(r0, r1)); j = int(rng.integers(c0, c1))
            if kind == "diag" and i == j: continue
            want = oracle.pairwise_block(np.vstack([freq[i:i + 1], freq[j:j + 1]]), metric, 0, 1)[0, 1]
            got = float(slab[i - lo, j])
            worst = max(worst, abs(got - want) / max(abs(want), 1e-300))
            if m is not None:
                assert float(m[j - c0, i - r0]) == got          # the mirror block holds the transposed entry
    print("rank %d: rows [%d,%d) %d blocks, %.1f ms (kernel %.1f), slab %.1f GB + mirrors %.1f GB, spot max rel err %.1e"
          % (g, lo, hi, len(mirrors), dt, st["kernel_ms"], slab.numel() * 8 / 1e9, sum(m.numel() for m in mirrors if m is not None) * 8 / 1e9, worst), flush=True)
    assert worst < 1e-6
    del slab, mirrors
    torch.cuda.empty_cache()
assert total_pairs == n * (n + 1) // 2, (total_pairs, n * (n + 1) // 2)
print("every unordered pair (and the diagonal) evaluated exactly once: %d; per-rank time %.1f..%.1f ms -> %.3e pairs/s on %d GPUs if they ran side by side"
      % (total_pairs, min(times), max(times), n * (n - 1) / 2 / (max(times) * 1e-3), world))

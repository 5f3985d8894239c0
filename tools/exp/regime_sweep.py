"""one-off: stage-2 wall time over record count and word-space size, every metric, uniform 2 kb contigs and ragged totals -
looking for cliffs (a regime where time per pair and word jumps)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
ctx = pa.Context(0)
rng = np.random.default_rng(1)


def run(n, k, metric, ragged):
    dim = 4 ** k
    lam = max(0.02, 4000.0 / dim)
    counts = rng.poisson(lam, size=(n, dim)).astype(np.uint32)
    if ragged:
        scale = rng.integers(1, 6, size=(n, 1))
        counts = (counts * scale).astype(np.uint32)
    totals = counts.sum(1).astype(np.uint64)
    dc, dt = torch.from_numpy(counts.view(np.int32)).cuda(), torch.from_numpy(totals.view(np.int64)).cuda()
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    best, kid = 1e9, None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, st = ctx.pairwise(dc, dt, metric, out=out, want_stats=True)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        kid = st["kernel_id"]
    return best, kid


which = sys.argv[1] if len(sys.argv) > 1 else "n"
if which == "n":
    for metric in ("Eucl", "JSD", "BC", "SC", "KT"):
        for ragged in (False, True):
            row = []
            for n in (64, 128, 512, 2048, 8192, 20000):
                t, kid = run(n, 4, metric, ragged)
                row.append("%6d: %8.3f ms (id %d, %6.2f ns/pair)" % (n, t * 1e3, kid, t * 1e9 / (n * n / 2)))
            print("%-4s %-7s k=4  " % (metric, "ragged" if ragged else "uniform") + " | ".join(row), flush=True)
else:
    n = 8192
    for metric in ("Eucl", "JSD", "BC", "SC", "KT"):
        for ragged in (False, True):
            row = []
            for k in (1, 2, 3, 4, 5, 6, 7):
                if metric == "KT" and k > 6:
                    continue
                t, kid = run(n if k < 7 else 4096, k, metric, ragged)
                nn = n if k < 7 else 4096
                row.append("k=%d: %8.2f ms (id %d, %6.3f ns/pair/word)" % (k, t * 1e3, kid, t * 1e9 / (nn * nn / 2) / 4 ** k))
            print("%-4s %-7s n=%d  " % (metric, "ragged" if ragged else "uniform", n) + " | ".join(row), flush=True)

"""Many calls through one context with changing sizes, metrics and entry points: buffers are reused or regrown, nothing
leaks, results stay finite.  Prints device memory in use at a few points."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
ctx = pa.Context(0)
rng = np.random.default_rng(0)
free0 = torch.cuda.mem_get_info()[0]
for it in range(300):
    n = int(rng.integers(1, 1500)); dim = int(rng.choice([4, 16, 64, 256, 1024]))
    counts = rng.integers(0, int(rng.choice([3, 100, 300, 30000])), size=(n, dim)).astype(np.uint32)
    totals = counts.sum(1).astype(np.uint64)
    metric = str(rng.choice(["Eucl", "JSD", "BC", "SC", "KT"]))
    if metric == "KT" and dim > 256 and n > 300:
        n = 300; counts = counts[:n]; totals = totals[:n]
    if rng.random() < 0.5:
        out = ctx.pairwise(counts, totals, metric, dtype=str(rng.choice(["float64", "float32"])))
    else:
        out = ctx.pairwise_freq(ctx.frequencies(counts, totals), metric)
    assert out.shape == (n, n)
    if it % 100 == 99:
        torch.cuda.synchronize()
        print("call %d: device memory in use by the process %.1f MB" % (it + 1, (free0 - torch.cuda.mem_get_info()[0]) / 1e6), flush=True)
ctx.close()
print("closed: %.1f MB" % ((free0 - torch.cuda.mem_get_info()[0]) / 1e6))

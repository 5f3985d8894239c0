"""one-off: Eucl at 50 000 records when some counts exceed two 7-bit digits (a few Mb-scale scaffolds in the assembly):
three-plane int8 Gram against the float64 Gram it used to fall back to (table_path=False)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
ctx = pa.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
rng = np.random.default_rng(3)
for top in (100, 10_000, 100_000, 2_000_000):
    counts = rng.integers(0, 60, size=(n, 256), dtype=np.uint32)
    counts[7] = rng.integers(0, top, size=256, dtype=np.uint32); counts[7, 0] = top
    totals = counts.sum(1).astype(np.uint64)
    dc, dt = torch.from_numpy(counts.view(np.int32)).cuda(), torch.from_numpy(totals.view(np.int64)).cuda()
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    res = {}
    for name, kw in (("int8 planes", {}), ("float64 Gram", {"table_path": False})):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.pairwise(dc, dt, "Eucl", out=out, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res[name] = (best, out[7, :2000].clone(), out[:2000, 7].clone())
    a, b = res["int8 planes"], res["float64 Gram"]
    print("largest count %9d: int8 planes %7.2f ms   float64 Gram %7.2f ms   max rel. difference on row 7: %.2e, symmetric %s"
          % (top, a[0] * 1e3, b[0] * 1e3, float(((a[1] - b[1]).abs() / b[1].clamp_min(1e-300)).max()), bool(torch.equal(a[1], a[2]))), flush=True)

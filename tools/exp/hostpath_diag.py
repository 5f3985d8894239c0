import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "|", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
n = 30000
ctx = pa.Context(0)
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
counts, totals = ctx.count_profiles(seq, off, "1111", "both")
for it in range(3):
    t = time.perf_counter(); out = np.zeros((n, n)); t0 = time.perf_counter() - t
    t = time.perf_counter(); ctx.pairwise(counts, totals, "Eucl", out=out); t1 = time.perf_counter() - t
    t = time.perf_counter(); ctx.pairwise(counts, totals, "Eucl", out=out); t2 = time.perf_counter() - t
    print("np.zeros %.1f ms | pairwise into fresh %.1f ms | again into the same (touched) %.1f ms" % (t0 * 1e3, t1 * 1e3, t2 * 1e3), flush=True)
    del out
out = np.empty((n, n))
t = time.perf_counter(); ctx.pairwise(counts, totals, "Eucl", out=out); print("np.empty fresh %.1f ms" % ((time.perf_counter() - t) * 1e3))
dc, dt = torch.from_numpy(counts.astype(np.int32)).cuda(), torch.from_numpy(totals.astype(np.int64)).cuda()
dout = torch.empty((n, n), dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter(); ctx.pairwise(dc, dt, "Eucl", out=dout); torch.cuda.synchronize(); print("device only %.1f ms" % ((time.perf_counter() - t) * 1e3))
pin = torch.empty((n, n), dtype=torch.float64).pin_memory()
torch.cuda.synchronize(); t = time.perf_counter(); pin.copy_(dout); torch.cuda.synchronize(); print("torch D2H into pinned %.1f ms" % ((time.perf_counter() - t) * 1e3))

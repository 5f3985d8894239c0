"""one-off: every metric at C2 with a float64 and with a float32 matrix (what --large memmap / h5py and the multi-GPU CLI ask for)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
n = 50000
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
counts, totals = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
outs = {"float64": torch.empty((n, n), dtype=torch.float64, device="cuda"), "float32": torch.empty((n, n), dtype=torch.float32, device="cuda")}
for metric in ("JSD", "Eucl", "BC", "SC", "KT"):
    row = []
    for name, out in outs.items():
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True, dtype=name)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        row.append("%s %6.2f ms" % (name, best * 1e3))
    print("%-4s (id %d)  " % (metric, st["kernel_id"]) + "   ".join(row), flush=True)

"""one-off: does the time of a store-bound matrix depend on WHERE the result buffer lies?  (A/B runs of one library showed two
modes per process: C2 float32 Eucl 2.00 or 2.18 ms, once float64 2.92 instead of 3.75.)  One process, the result tensor allocated
again and again (allocator cache emptied in between, a dummy of varying size before it): address, best-of-3 kernel time."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
n = 50000
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
rng = np.random.default_rng(5)
for dt in (torch.float32, torch.float64):
    for trial in range(10):
        torch.cuda.empty_cache()
        dummy = torch.empty(int(rng.integers(1, 3000)) * (1 << 20), dtype=torch.uint8, device="cuda") if trial else None
        out = torch.empty((n, n), dtype=dt, device="cuda")
        ts = []
        for _ in range(4):
            _, st = ctx.pairwise(c, t, "Eucl", out=out, want_stats=True, dtype=dt)
            ts.append(st["kernel_ms"])
        print("%-8s trial %2d  ptr 0x%x (mod 2 MiB %7d, mod 1 GiB %5d MiB)  kernel best %.3f  all %s" % (
            str(dt).replace("torch.", ""), trial, out.data_ptr(), out.data_ptr() % (2 << 20), (out.data_ptr() % (1 << 30)) >> 20,
            min(ts), " ".join("%.2f" % x for x in ts)), flush=True)
        del out, dummy

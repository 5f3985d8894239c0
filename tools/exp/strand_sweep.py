"""one-off: stage 2 at C2 size for -s both / plus and a pattern that is not its own mirror image (no reverse-complement folding)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
n = 50000
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
dseq, doff = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
for pattern, strand in (("1111", "both"), ("1111", "plus"), ("1111", "minus"), ("1101", "both"), ("11011", "both"), ("111", "both"), ("11111", "both")):
    counts, totals = ctx.count_profiles(dseq, doff, pattern, strand)
    row = []
    for metric in ("Eucl", "JSD", "BC", "SC", "KT"):
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        row.append("%s %6.2f ms (id %d%s)" % (metric, best * 1e3, st["kernel_id"], ", folded" if st["rc_folded"] else ""))
    print("%-6s %-5s D=%4d  " % (pattern, strand, counts.shape[1]) + " | ".join(row), flush=True)

"""one-off: the frequency-matrix entry point (po_pairwise_freq: what a ctypes stub inside the reference calls) against the count
entry point on the ragged assembly and on C2: does the integer recovery hold at real totals, and what does it cost?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
for name, (seq, off) in (("C2", synthetic.contig_bytes(50000, 2000, seed=50001)), ("ragged", synthetic.ragged_assembly(50000, seed=2024)[:2])):
    dseq, doff = torch.from_numpy(np.asarray(seq)).cuda(), torch.from_numpy(np.asarray(off).astype(np.int64)).cuda()
    counts, totals = ctx.count_profiles(dseq, doff, "1111", "both")
    freq = ctx.frequencies(counts, totals)           # device float64 [n, 256]
    n = counts.shape[0]
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    for metric in ("Eucl", "JSD", "BC", "SC", "KT"):
        res = []
        for label, fn in (("counts", lambda: ctx.pairwise(counts, totals, metric, out=out, want_stats=True)),
                          ("freq", lambda: ctx.pairwise_freq(freq, metric, out=out, want_stats=True))):
            best, st = 1e9, None
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _, st = fn()
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            res.append((label, best, st["kernel_id"], out[123, :4000].clone()))
        same = bool(torch.equal(res[0][3], res[1][3]))
        print("%-6s %-4s  counts: %7.2f ms (id %d)   freq: %7.2f ms (id %d)   row 123 identical: %s" % (name, metric, res[0][1] * 1e3, res[0][2], res[1][1] * 1e3, res[1][2], same), flush=True)

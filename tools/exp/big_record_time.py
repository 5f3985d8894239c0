"""one-off: stage-1 time as a function of record length (same bytes, cut differently): the flush of a record that spans
many workgroups goes through global atomics on ONE row of the count matrix."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import phyloligo_amd as pa

total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
pattern = sys.argv[2] if len(sys.argv) > 2 else "1111"
ctx = pa.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(5)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
seq = lut[torch.randint(0, 4, (total,), device="cuda", generator=g)]
sizes = [int(float(x)) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [2_000, 20_000, 200_000, 2_000_000, 20_000_000, 200_000_000, total]
for rec_len in sizes:
    n = max(1, total // rec_len)
    off = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * rec_len
    off[-1] = total
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        counts, totals = ctx.count_profiles(seq, off, pattern, "both")
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("records of %11d bytes (%7d records): %8.3f ms  %6.2f TB/s of sequence" % (rec_len, n, best * 1e3, total / best / 1e12), flush=True)
    del counts, totals

"""one-off: stage 1 on the ragged 0.33 Gb assembly with more and more dirt: lower case everywhere, an N every 10 000 / 1 000 / 100 bases,
half of every record one long N run"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
seq, off = synthetic.ragged_assembly(50000, seed=2024, dirt=False)[:2]
off64 = np.asarray(off).astype(np.int64)
doff = torch.from_numpy(off64).cuda()


def best(dseq, pattern="1111", strand="both", reps=5):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        c, tt = ctx.count_profiles(dseq, doff, pattern, strand)
        torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    return b, int(tt.sum())


base = torch.from_numpy(np.asarray(seq)).cuda()
cases = [("clean upper case", base)]
cases.append(("all lower case", base | 0x20))
for every in (10000, 1000, 100):
    s = base.clone(); s[::every] = ord("N"); cases.append(("an N every %d bases" % every, s))
s = base.clone()
mid = torch.from_numpy((off64[:-1] + off64[1:]) // 2).cuda()
mask = torch.zeros(base.numel() + 1, dtype=torch.int32, device="cuda")
mask[doff[:-1]] += 1; mask[mid] -= 1
s[torch.cumsum(mask[:-1], 0) > 0] = ord("N")                       # the first half of every record
cases.append(("first half of every record N", s))
s = base.clone(); s[::7] = ord("R"); cases.append(("an IUPAC code every 7 bases", s))
for name, s in cases:
    for pattern in ("1111", "11011011"):
        t, words = best(s, pattern)
        print("%-32s %-9s %8.1f us  %5.2f TB/s of sequence   words counted %d" % (name, pattern, t * 1e6, base.numel() / t / 1e12, words), flush=True)

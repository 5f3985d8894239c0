"""one-off: stage 1 on more than 4 GiB of sequence (byte offsets beyond 2^32, one record beyond 2^31 bytes).
Layout: [probe records] [filler: one 2.5 GB record + 2 000 records of ~1 MB] [the probe records again]; the probe set is
small (oracle-checkable), dirty and ragged.  Checks: probes at both ends == oracle, filler totals, row sums, the big record against a torch histogram (plus strand)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
from oracle import phyloligo_oracle as oracle

pattern, strand = (sys.argv[1] if len(sys.argv) > 1 else "1111"), (sys.argv[2] if len(sys.argv) > 2 else "both")
big = int(float(sys.argv[3])) if len(sys.argv) > 3 else 2_500_000_000
ctx = pa.Context(0)
pseq, poff = synthetic.ragged_assembly(n=48, seed=7, median=20000, sigma=1.0, lo=100, hi=300000)[:2]
pseq = np.asarray(pseq); poff = np.asarray(poff, dtype=np.int64)
print("probe: %d records, %d bytes" % (len(poff) - 1, poff[-1]), flush=True)
rng = np.random.default_rng(11)
fill_lens = np.concatenate([[big], rng.integers(500_000, 1_500_000, size=2000)]).astype(np.int64)
g = torch.Generator(device="cuda"); g.manual_seed(5)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
total_fill = int(fill_lens.sum())
filler = torch.empty(total_fill, dtype=torch.uint8, device="cuda")
step = 1 << 30
for a in range(0, total_fill, step):
    b = min(total_fill, a + step)
    filler[a:b] = lut[torch.randint(0, 4, (b - a,), device="cuda", generator=g)]
p = torch.from_numpy(pseq).cuda()
seq = torch.cat([p, filler, p]); del filler
lens = np.concatenate([np.diff(poff), fill_lens, np.diff(poff)])
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
assert off[-1] == seq.numel()
print("sequence: %.2f GB, %d records, longest %.2f GB; second probe set starts at byte %d (2^32 = %d)"
      % (off[-1] / 1e9, len(lens), lens.max() / 1e9, off[-1 - (len(poff) - 1)], 1 << 32), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
counts, totals = ctx.count_profiles(seq, torch.from_numpy(off).cuda(), pattern, strand)
torch.cuda.synchronize(); print("count_profiles %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
np_ = len(poff) - 1
c = counts.cpu().numpy().view(np.uint32).astype(np.int64); t = totals.cpu().numpy().astype(np.int64)
oc, ot = oracle.compute_counts([pseq[poff[i]:poff[i + 1]].tobytes() for i in range(np_)], pattern, strand)
ok1 = np.array_equal(c[:np_], oc) and np.array_equal(t[:np_], ot)
ok2 = np.array_equal(c[-np_:], oc) and np.array_equal(t[-np_:], ot)
span = len(pattern)
# 'both' = the record followed by its reverse complement as ONE string (select_strand, phyloligo.py:141): 2 L - span + 1 windows
exp_tot = np.maximum((fill_lens * 2 if strand == "both" else fill_lens) - span + 1, 0)
ok3 = np.array_equal(t[np_:-np_], exp_tot)
ok4 = np.array_equal(c.sum(axis=1), t)
print("probes at the start == oracle:", ok1)
print("probes beyond 4 GiB == oracle:", ok2)
print("filler totals (incl. the %.2f GB record):" % (big / 1e9), ok3, "" if ok3 else (t[np_:np_ + 3], exp_tot[:3]))
print("row sums == totals:", ok4)
# the big record's counts against a float64 histogram of its words computed by torch in pieces (plain k-mers only)
if set(pattern) == {"1"} and strand == "plus" and span <= 4:
    a0 = int(off[np_]); L = int(fill_lens[0])
    code = torch.zeros(256, dtype=torch.int64, device="cuda")
    for i, ch in enumerate(b"CGAT"):
        code[ch] = i
    hist = torch.zeros(4 ** span, dtype=torch.int64, device="cuda")
    piece = 1 << 28
    for s in range(0, L - span + 1, piece):
        e = min(L - span + 1, s + piece)
        w = torch.zeros(e - s, dtype=torch.int64, device="cuda")
        for j in range(span):
            w = w * 4 + code[seq[a0 + s + j:a0 + e + j].long()]
        hist += torch.bincount(w, minlength=4 ** span)
    print("big record == torch histogram:", bool(torch.equal(hist.cpu(), torch.from_numpy(c[np_]))))
print("ALL OK" if (ok1 and ok2 and ok3 and ok4) else "MISMATCH")

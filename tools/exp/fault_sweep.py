"""one-off: every metric x k = 1..6 x strand x output type at record counts around the tile edge and the 8 192-record switch -
looking for faults and wrong results in corners no test sits in (round 5: this kind of sweep found the k = 2 fold overflow).
Checks per case: exactly symmetric, the diagonal, float32 == float64 rounded once (where the path promises it: Eucl, SC, BC) or
within 2e-6, and two rows against the oracle (not Kendall beyond k = 4: O(D^2) per pair)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
from oracle import phyloligo_oracle as oracle
ctx = pa.Context(0)
bad = 0
t_start = time.time()
for k in (1, 2, 3, 4, 5, 6):
    for n in (127, 129, 8191, 8193, 12345):
        for strand in ("both", "plus"):
            seq, off = synthetic.contig_bytes(n, 1500 + 37 * k, seed=1000 + k)
            counts, totals = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1" * k, strand)
            ch, th = counts.cpu().numpy().view(np.uint32), totals.cpu().numpy().view(np.uint64)
            freq = oracle.counts_to_frequencies(ch.astype(np.int64), th.astype(np.int64))
            for metric in ("Eucl", "JSD", "BC", "SC", "KT"):
                if metric == "KT" and k == 6 and n > 8193:
                    continue
                t0 = time.time()
                o64 = torch.full((n, n), float("nan"), dtype=torch.float64, device="cuda")
                o32 = torch.full((n, n), float("nan"), dtype=torch.float32, device="cuda")
                ctx.pairwise(counts, totals, metric, out=o64)
                ctx.pairwise(counts, totals, metric, out=o32, dtype="float32")
                torch.cuda.synchronize()
                msgs = []
                if bool(torch.isnan(o64).any()) or bool(torch.isnan(o32).any()):
                    msgs.append("NaN / unwritten entries")
                if not bool(torch.equal(o64, o64.T)) or not bool(torch.equal(o32, o32.T)):
                    msgs.append("not symmetric")
                d = torch.diagonal(o64)
                if not bool((d == (1.0 if metric == "KT" else 0.0)).all()):
                    msgs.append("diagonal")
                if metric in ("Eucl", "SC", "BC"):
                    if not bool(torch.equal(o32, o64.to(torch.float32))):
                        msgs.append("float32 != rounded float64 (%d entries)" % int((o32 != o64.to(torch.float32)).sum()))
                elif not bool(torch.allclose(o32.double(), o64, rtol=2e-6, atol=1e-6)):
                    msgs.append("float32 far from float64")
                if not (metric == "KT" and k > 4):
                    rows = [1, n - 2]
                    want = oracle.pairwise_rows(freq, metric, rows)
                    got = o64[rows].cpu().numpy()
                    if not np.allclose(got, want, rtol=1e-6, atol=1e-9 if metric in ("SC", "KT") else 1e-12):
                        msgs.append("oracle rows differ by %.2e" % float(np.abs(got - want).max()))
                del o64, o32
                if msgs:
                    bad += 1
                    print("BAD k=%d n=%d %s %s: %s" % (k, n, strand, metric, "; ".join(msgs)), flush=True)
        print("k=%d n=%d done (%.0f s)" % (k, n, time.time() - t_start), flush=True)
print("cases with findings:", bad, flush=True)

"""one-off: how good are JSD / BC / SC / KT for NEAR-IDENTICAL records (one to three counts moved by 1 .. 3)?  Relative error of the
near pairs against an 80-bit evaluation of the reference's formulas, for the equal-total kernels and the general ones."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from oracle import phyloligo_oracle as oracle
ctx = pa.Context(0)
L = np.longdouble


def jsd_ld(p, q):
    p, q = p.astype(L), q.astype(L)
    h = (p + q) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(p > 0, p * np.log(p / h), 0)
        b = np.where(q > 0, q * np.log(q / h), 0)
    return float((a.sum() + b.sum()) / 2)


def bc_ld(p, q):
    p, q = p.astype(L), q.astype(L)
    return float(np.abs(p - q).sum() / (p + q).sum())


for top, dim, equal in ((60, 256, True), (60, 256, False), (5000, 256, False), (200_000, 256, False), (120, 4096, True)):
    rng = np.random.default_rng(top + dim)
    n = 300
    counts = rng.integers(top // 4, top // 2, size=(n, dim)).astype(np.int64)
    near = []
    for t in range(40):
        a, b = 10 + 2 * t, 11 + 2 * t
        counts[b] = counts[a]
        ws = rng.choice(dim, size=2 * (1 + t % 3), replace=False)
        for i, w in enumerate(ws):                         # moved, not added: the totals stay equal
            counts[b, w] += (1 + t % 3) * (1 if i % 2 == 0 else -1)
        near.append((a, b))
    if equal:
        target = counts.sum(1).max()
        counts[:, 0] += target - counts.sum(1)
        for a, b in near:
            counts[b, 0] = counts[a, 0]
    else:
        counts[::2, 1] += rng.integers(0, top // 4 + 1, size=counts[::2, 1].shape)
        for a, b in near:
            counts[b, 1] = counts[a, 1]
    totals = counts.sum(1)
    dc, dt = torch.from_numpy(counts.astype(np.int32)).cuda(), torch.from_numpy(totals).cuda()
    fl = counts.astype(L) / totals.astype(L)[:, None]
    for metric, ref in (("JSD", jsd_ld), ("BC", bc_ld)):
        for name, kw in (("default", {}), ("general", {"table_path": False, "rc_fold": False})):
            got, st = ctx.pairwise(dc, dt, metric, want_stats=True, **kw)
            got = got.cpu().numpy()
            errs = [abs(got[a, b] - ref(fl[a], fl[b])) / ref(fl[a], fl[b]) for a, b in near]
            vals = [ref(fl[a], fl[b]) for a, b in near]
            print("top %7d dim %4d equal %-5s %-3s %-7s kernel %d   values %.1e .. %.1e   relative error max %.1e median %.1e" % (
                top, dim, equal, metric, name, st["kernel_id"], min(vals), max(vals), max(errs), float(np.median(errs))), flush=True)

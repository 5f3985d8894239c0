"""one-off: an ADVERSARIAL fuzz for relative accuracy - records that are perturbations of each other by anything from one k-mer to ten
per cent, equal and unequal totals, every plane class, counts and frequency entry points - Eucl / BC / JSD against references evaluated
from the integers without cancellation (Eucl, BC: exact rational arithmetic; JSD: the phi form in 80-bit arithmetic).  rtol 1e-6, no atol."""
import sys, math
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
ctx = pa.Context(0)
L = np.longdouble
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0


def refs(a, na, b, nb):
    a, b = [int(x) for x in a], [int(x) for x in b]
    na, nb = int(na), int(nb)
    x = [ak * nb for ak in a]
    y = [bk * na for bk in b]
    num = sum((xi - yi) ** 2 for xi, yi in zip(x, y))
    eucl = math.sqrt(num) / (na * nb) if num < 2 ** 1000 else float("nan")
    if num >= 2 ** 53:                                   # sqrt of a big integer: via isqrt with extra digits
        eucl = math.isqrt(num << 120) / (2 ** 60) / (na * nb)
    sad, tot = sum(abs(xi - yi) for xi, yi in zip(x, y)), sum(xi + yi for xi, yi in zip(x, y))
    bc = sad / tot if tot else float("nan")
    acc = L(0)
    for xi, yi in zip(x, y):
        if xi + yi == 0:
            continue
        t = L(xi - yi) / L(xi + yi)
        u = t * t
        if u <= L(1) / 16:
            phi = sum(u ** i / (i * (2 * i - 1)) for i in range(1, 24))
        elif xi == 0 or yi == 0:
            phi = 2 * np.log(L(2))
        else:
            phi = (1 + t) * np.log(1 + t) + (1 - t) * np.log(1 - t)
        acc += L(xi + yi) * phi
    jsd = float(acc / (4 * L(na) * L(nb)))
    return {"Eucl": eucl, "BC": bc, "JSD": jsd}


for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([4, 16, 64, 256, 1024]))
    top = int(rng.choice([8, 60, 127, 300, 5000, 20000, 300_000, 2_000_000, 2_500_000]))
    n = int(rng.choice([130, 300]))
    counts = rng.integers(0, top + 1, size=(n, dim)).astype(np.int64)
    if rng.random() < 0.3:
        counts[rng.random((n, dim)) < 0.5] = 0
    pairs = []
    for t in range(30):
        a, b = 2 * t, 2 * t + 1
        counts[b] = counts[a]
        mode = rng.integers(0, 4)
        k = int(rng.integers(1, max(2, dim // 2)))
        ws = rng.choice(dim, size=min(dim, k), replace=False)
        if mode == 0:                                    # a few k-mers added
            counts[b, ws[:3]] += rng.integers(1, 4, size=len(ws[:3]))
        elif mode == 1:                                  # moved (totals equal)
            for i, w in enumerate(ws[:4]):
                counts[b, w] = max(0, counts[b, w] + (1 if i % 2 == 0 else -1))
        elif mode == 2:                                  # scaled copy plus noise: proportional up to a few counts
            counts[b] = counts[a] * int(rng.integers(2, 4))
            counts[b, ws[:2]] += 1
        else:                                            # up to 10 per cent noise
            counts[b, ws] = (counts[b, ws] * (1 + 0.1 * rng.random(len(ws)))).astype(np.int64)
        pairs.append((a, b))
    if rng.random() < 0.4:
        target = counts.sum(1).max()
        counts[:, 0] += target - counts.sum(1)
    counts = np.minimum(counts, 2 ** 31 - 1)
    totals = counts.sum(1)
    if totals.max() >= 2 ** 32 or (totals == 0).any():
        continue
    dc, dt = torch.from_numpy(counts.astype(np.int32)).cuda(), torch.from_numpy(totals).cuda()
    freq = counts / totals[:, None]
    for metric in ("Eucl", "BC", "JSD"):
        res = {"counts": ctx.pairwise(dc, dt, metric).cpu().numpy(), "general": ctx.pairwise(dc, dt, metric, table_path=False, rc_fold=False).cpu().numpy(),
               "freq": ctx.pairwise_freq(freq, metric)}
        for a, b in pairs:
            want = refs(counts[a], totals[a], counts[b], totals[b])[metric]
            for name, got in res.items():
                g = got[a, b]
                # frequencies that could not be turned back into integers carry the rounding of the quotients: 1e-16 / relative difference
                tol = 1e-6
                ok = (g == want) or (want != 0 and abs(g - want) <= tol * abs(want)) or (math.isnan(want) and math.isnan(g))
                if not ok:
                    bad += 1
                    print("seed %d dim %d top %d %s %s pair (%d,%d): got %.17g want %.17g rel %.2e" % (
                        seed, dim, top, metric, name, a, b, g, want, abs(g - want) / abs(want) if want else float("inf")), flush=True)
    if (seed - lo) % 20 == 0:
        print("  seed %d done, findings so far %d" % (seed, bad), flush=True)
print("findings:", bad, flush=True)

"""JSD of near-identical records (csrc/po_jsd_exact.hip).  The tile kernels' 1/2 (E_a + E_b - S) is good to ~1e-14 absolute; records
that differ by a handful of k-mers have JSD 1e-8 ... 1e-13, where that is 1e-6 ... 1e-2 in relative terms, and the reference's own
per-word form (phylodist.py:18-24, :43-48) is relatively accurate.  Values below 2^-20 are therefore evaluated again word by word in
a cancellation-free form.  Checked against the reference's formula in 80-bit arithmetic and against a cancellation-free evaluation from the integers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L = np.longdouble


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def jsd_80bit(p, q):
    p, q = p.astype(L), q.astype(L)
    h = (p + q) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(p > 0, p * np.log(p / h), 0)
        b = np.where(q > 0, q * np.log(q / h), 0)
    return float((a.sum() + b.sum()) / 2)


def jsd_phi_form(a, na, b, nb):
    """the same quantity without cancellation, from the integers: 1/2 sum_k m_k phi(t_k), t = (x - y) / (x + y), phi(t) = (1 + t) ln(1 + t)
    + (1 - t) ln(1 - t) = sum_i t^(2i) / (i (2i - 1)); t and m exactly from Python integers, the rest in 80-bit arithmetic"""
    acc = L(0)
    for ak, bk in zip(a.tolist(), b.tolist()):
        x, y = ak * int(nb), bk * int(na)
        if x + y == 0:
            continue
        t = L(x - y) / L(x + y)
        u = t * t
        if u <= L(1) / 16:
            phi = sum(u ** i / (i * (2 * i - 1)) for i in range(1, 20))
        else:
            phi = (1 + t) * np.log(1 + t) + ((1 - t) * np.log(1 - t) if t < 1 else 0)
        acc += L(x + y) * phi
    return float(acc / (4 * L(int(na)) * L(int(nb))))


def near_identical_counts(n, dim, top, equal_totals, seed):
    """records 10 + 2 t + 1 = record 10 + 2 t with one to three k-mers MOVED (totals unchanged); record 5 = record 4 exactly"""
    rng = np.random.default_rng(seed)
    counts = rng.integers(top // 4, top // 2, size=(n, dim)).astype(np.int64)
    near = []
    for t in range(40):
        a, b = 10 + 2 * t, 11 + 2 * t
        counts[b] = counts[a]
        ws = rng.choice(dim, size=2 * (1 + t % 3), replace=False)
        for i, w in enumerate(ws):
            counts[b, w] += (1 + t % 3) * (1 if i % 2 == 0 else -1)
        near.append((a, b))
    if equal_totals:
        target = counts.sum(1).max()
        counts[:, 0] += target - counts.sum(1)
    else:
        counts[::2, 1] += rng.integers(0, top // 4 + 1, size=counts[::2, 1].shape)
    for a, b in near:
        counts[b, 0], counts[b, 1] = counts[a, 0], counts[a, 1]
    counts[5] = counts[4]
    return counts, counts.sum(1), near


@pytest.mark.parametrize("top,dim,equal_totals", [(60, 256, True), (60, 256, False), (5000, 256, False), (200_000, 256, False),
                                                  (120, 4096, True), (200_000, 64, True)])
def test_jsd_of_near_identical_records(ctx, top, dim, equal_totals):
    import torch
    n = 300
    counts, totals, near = near_identical_counts(n, dim, top, equal_totals, seed=top + dim)
    dc, dt = torch.from_numpy(counts.astype(np.int32)).cuda(), torch.from_numpy(totals).cuda()
    fl = counts.astype(L) / totals.astype(L)[:, None]
    freq = (counts / totals[:, None]).astype(np.float64)
    want = {ab: jsd_phi_form(counts[ab[0]], totals[ab[0]], counts[ab[1]], totals[ab[1]]) for ab in near}
    for ab, w in want.items():        # the reference's own form in 80-bit arithmetic agrees as far as ITS cancellation allows (first-order terms cancel)
        assert abs(jsd_80bit(fl[ab[0]], fl[ab[1]]) - w) <= 1e-7 * w
    assert 0 < min(want.values()) and max(want.values()) < 1e-4
    assert top <= 120 or max(want.values()) < 2.0 ** -20                   # below 2^-20: evaluated again; just above (top = 60): 1e-7 as it is
    results = {"counts": ctx.pairwise(dc, dt, "JSD").cpu().numpy(),
               "general kernel": ctx.pairwise(dc, dt, "JSD", table_path=False, rc_fold=False).cpu().numpy(),
               "frequencies": ctx.pairwise_freq(freq, "JSD"),
               "frequencies, general kernel": ctx.pairwise_freq(freq, "JSD", table_path=False)}
    for name, got in results.items():
        assert np.array_equal(got, got.T), name
        assert got[4, 5] == 0.0 and got[5, 4] == 0.0, name                 # identical records: exactly 0, as the reference gives
        for (a, b), w in want.items():
            # (from float64 frequencies the differences x - y carry the rounding of the quotients: 1e-16 / t)
            tol = 3e-7 if w >= 0.9 * 2.0 ** -20 else (1e-9 if name == "frequencies, general kernel" else 1e-12)
            assert abs(got[a, b] - w) <= tol * w, (name, a, b, got[a, b], w)
    f32 = ctx.pairwise(dc, dt, "JSD", dtype="float32").cpu().numpy()
    for (a, b), w in want.items():
        assert f32[a, b] == np.float32(results["counts"][a, b]) and f32[b, a] == f32[a, b]
    # a block with a mirror buffer, and a row range, give the same bits as the full matrix
    out = torch.full((120, 200), float("nan"), dtype=torch.float64, device="cuda")
    mir = torch.full((200, 120), float("nan"), dtype=torch.float64, device="cuda")
    ctx.pairwise_blocks(dc, dt, "JSD", [{"rows": (3, 123), "cols": (7, 207), "out": out, "mirror": mir}])
    full = results["counts"]
    assert np.array_equal(out.cpu().numpy(), full[3:123, 7:207]) and np.array_equal(mir.cpu().numpy(), full[7:207, 3:123])
    rows = ctx.pairwise(dc, dt, "JSD", row_begin=9, row_end=140).cpu().numpy()
    assert np.array_equal(rows, full[9:140])


def test_more_flagged_rows_than_the_list_holds(ctx):
    """6 000 copies of one record: every row of every tile is noted (17 000 entries, the list holds 16 384), so none is evaluated
    again - all or nothing - and the values stay what the tile kernel gave: rounding level, symmetric."""
    import torch
    rng = np.random.default_rng(1)
    base = rng.integers(5, 40, size=64).astype(np.int32)
    counts = np.tile(base, (6000, 1))
    totals = counts.astype(np.int64).sum(1)
    got = ctx.pairwise(torch.from_numpy(counts).cuda(), torch.from_numpy(totals).cuda(), "JSD").cpu().numpy()
    assert np.array_equal(got, got.T) and float(np.abs(got).max()) < 5e-14

"""Seeded fuzz of stage 2 through the C ABI: random sizes, dimensions (powers of 4 and not), count ranges that hit
every kernel family, strand-symmetric and asymmetric records, equal and ragged totals, row ranges, output types,
count and frequency entry points - every result against the oracle (1e-6 relative, as the north star asks)."""
import numpy as np
import pytest

from oracle import phyloligo_oracle as oracle

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-12


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def rc_perm(dim):
    k = int(round(np.log(dim) / np.log(4)))
    w = np.arange(dim)
    r = np.zeros(dim, dtype=np.int64)
    x = w.copy()
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 1)
        x >>= 2
    return r


def make_case(rng):
    n = int(rng.choice([1, 2, 5, 64, 127, 128, 129, 200, 257, 300]))
    dim = int(rng.choice([1, 3, 4, 16, 50, 64, 256]))
    top = int(rng.choice([1, 3, 40, 127, 128, 200, 255, 256, 20000, 1_500_000]))     # ... 200: wide JSD table; 1.5e6: three int8 planes
    counts = rng.integers(0, top + 1, size=(n, dim)).astype(np.uint32)
    counts[rng.random((n, dim)) < rng.choice([0.0, 0.3, 0.9])] = 0
    if rng.random() < 0.2:                                       # no rare word (a long contig): the frequency entry point has to
        counts = counts + np.uint32(rng.integers(256, 2000))     # find the total by continued fractions
    if dim in (4, 16, 64, 256) and rng.random() < 0.5:          # strand-symmetric records
        counts = counts + counts[:, rc_perm(dim)]
    if rng.random() < 0.5 and n > 2:                             # equal totals: top up one word per record
        target = int(counts.sum(1).max())
        counts[:, 0] += (target - counts.sum(1)).astype(np.uint32)
        if dim in (4, 16, 64, 256) and rng.random() < 0.5:
            pass
    if n > 3:
        counts[1] = counts[0]                                    # duplicate
        if rng.random() < 0.5:
            counts[2] = 0                                        # empty record
    totals = counts.sum(1).astype(np.uint64)
    return counts, totals


@pytest.mark.parametrize("seed", range(60))
def test_random_problem(ctx, seed):
    rng = np.random.default_rng(7000 + seed)
    counts, totals = make_case(rng)
    n, dim = counts.shape
    metric = str(rng.choice(["Eucl", "JSD", "BC", "SC", "KT"]))
    freq = oracle.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
    want = oracle.pairwise_block(freq, metric)
    atol = 1e-9 if metric == "SC" else ATOL                     # 1 - r near 1 loses absolute, not relative, accuracy
    lo = int(rng.integers(0, n))
    hi = int(rng.integers(lo, n + 1))
    variants = [
        ("counts", lambda: ctx.pairwise(counts, totals, metric)),
        ("general", lambda: ctx.pairwise(counts, totals, metric, table_path=False, rc_fold=False)),
        ("nosym", lambda: ctx.pairwise(counts, totals, metric, symmetric=False)),
        ("freq", lambda: ctx.pairwise_freq(freq, metric)),
        ("freq-general", lambda: ctx.pairwise_freq(freq, metric, table_path=False)),
    ]
    for name, fn in variants:
        got = fn()
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=atol, equal_nan=True, err_msg="%s %s n=%d dim=%d" % (name, metric, n, dim))
        assert np.array_equal(np.isnan(got), np.isnan(want)), (name, metric)
    rows = ctx.pairwise(counts, totals, metric, row_begin=lo, row_end=hi)
    np.testing.assert_allclose(rows, want[lo:hi], rtol=RTOL, atol=atol, equal_nan=True)
    f32 = ctx.pairwise(counts, totals, metric, dtype="float32")
    np.testing.assert_allclose(f32, want.astype(np.float32), rtol=2e-6, atol=1e-6, equal_nan=True)

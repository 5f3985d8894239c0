"""Exact int8-MFMA Gram kernels (csrc/po_gram_i8.hip): Euclidean distance with one digit plane (counts <= 127),
two digit planes (counts <= 16 383), three (counts <= 2 097 151: scaffolds and chromosomes) and the float64 kernel beyond
that; Spearman from doubled centred ranks."""
import numpy as np
import pytest

from oracle import phyloligo_oracle as oracle

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-12
I8, F64 = 4, 3           # PO_KERNEL_MFMA_I8_GRAM, PO_KERNEL_MFMA_F64_GRAM


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def random_counts(n, dim, top, seed, empty=(3,), dup=((5, 17),)):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, max(2, top // 3), size=(n, dim), dtype=np.uint32)
    counts[rng.random((n, dim)) < 0.2] = 0
    counts[1, dim // 2] = top                       # the largest value decides which kernel runs
    for e in empty:
        counts[e] = 0
    for a, b in dup:
        counts[b] = counts[a]
    return counts, counts.sum(1).astype(np.uint64)


@pytest.mark.parametrize("top,kernel", [(127, I8), (128, I8), (5000, I8), (16383, I8), (16384, I8), (100_000, I8), (2_097_151, I8),
                                        (2_097_152, F64), (3_000_000, F64)])
@pytest.mark.parametrize("dim", [16, 256, 200])
def test_eucl_digit_planes(ctx, top, kernel, dim):
    counts, totals = random_counts(300, dim, top, seed=top + dim)
    got, st = ctx.pairwise(counts, totals, "Eucl", want_stats=True)
    assert st["kernel_id"] == I8                    # both kernel families are launched; the device flag picks one
    freq = oracle.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
    want = oracle.pairwise_block(freq, "Eucl")
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
    assert np.array_equal(got, got.T) and np.all(np.diag(got) == 0.0)
    assert got[5, 17] == 0.0                        # duplicates: exactly 0, in every kernel
    general = ctx.pairwise(counts, totals, "Eucl", table_path=False)
    np.testing.assert_allclose(general, want, rtol=RTOL, atol=ATOL)
    if kernel == I8 and top > 127:
        # two / three planes: G is an exact integer, so the result does not depend on tiling or on the row range
        sub = ctx.pairwise(counts, totals, "Eucl", row_begin=40, row_end=171)
        assert np.array_equal(sub, got[40:171])
        f32 = ctx.pairwise(counts, totals, "Eucl", dtype="float32")
        assert np.array_equal(f32, got.astype(np.float32))


@pytest.mark.parametrize("top,dim", [(100, 4096), (5000, 256), (200_000, 256), (2_000_000, 64), (3_000_000, 64)])
@pytest.mark.parametrize("n", [300, 8300])
def test_eucl_near_duplicates_keep_their_relative_accuracy(ctx, top, dim, n):
    """Records that differ from another by one or a few k-mers: in  S_a/n_a^2 + S_b/n_b^2 - 2 G/(n_a n_b)  the three rounded terms
    cancel to d^2 / sum = 1e-7 ... 1e-13, and the square root of that was good to 1e-6 ... 1e-5 only (a fuzz seed at the end of
    round 5: two near-identical records, error 2e-6).  From 2^-21 on the squared distance now comes from the exact integers in
    double-double arithmetic; the float64 Gram kernel (counts beyond 2 097 151, and table_path=False) sums the squared differences
    of such a pair word by word, as the reference does.  One, two and three digit planes and the float64 kernel, both sides of the
    8 192-record switch, float32 = rounded float64, exactly symmetric; against the oracle's direct sum of squared differences
    (itself good to ~1e-11 here)."""
    import torch
    rng = np.random.default_rng(top + dim + n)
    counts = rng.integers(top // 4, top // 2, size=(n, dim)).astype(np.uint32)
    counts[0, 0] = top                                       # the plane class of the whole matrix (below 8 192 records) / of block 0
    near = []
    for t in range(40):                                      # record 10 + 2 t + 1 = record 10 + 2 t with one to three counts moved by 1 .. 3
        a, b = 10 + 2 * t, 11 + 2 * t
        counts[b] = counts[a]
        for w in rng.integers(0, dim, size=1 + t % 3):
            counts[b, w] += 1 + t % 3
        near.append((a, b))
    totals = counts.astype(np.int64).sum(1)
    dc, dt = torch.from_numpy(counts.view(np.int32)).cuda(), torch.from_numpy(totals).cuda()
    got = ctx.pairwise(dc, dt, "Eucl").cpu().numpy()
    f32 = ctx.pairwise(dc, dt, "Eucl", dtype="float32").cpu().numpy()
    assert np.array_equal(got, got.T) and np.array_equal(f32, got.astype(np.float32))
    freq = oracle.counts_to_frequencies(counts.astype(np.int64), totals)
    want = oracle.pairwise_block(freq, "Eucl", row_begin=10, row_end=90)      # the near-duplicate records against everybody
    np.testing.assert_allclose(got[10:90], want, rtol=1e-9, atol=1e-300)
    if n <= 300:                                             # the same through the float64 Gram kernel
        general = ctx.pairwise(dc, dt, "Eucl", table_path=False).cpu().numpy()
        np.testing.assert_allclose(general[10:90], want, rtol=1e-7, atol=1e-300)      # (its Gram entries are rounded sums of D products)
        assert np.array_equal(general, general.T)
    ratios = [got[a, b] ** 2 / ((freq[a] ** 2).sum() + (freq[b] ** 2).sum()) for a, b in near]
    assert min(ratios) > 0 and min(ratios) < 2.0 ** -22 and max(ratios) < 1e-5   # the pairs the test is about are at cancellation level


def test_eucl_distance_far_below_the_resolution_of_the_gram_form(ctx):
    """One word holds most of every record (5e8 of 8e8 k-mers - beyond the int8 planes: the float64 Gram kernel) and two records
    differ by two k-mers: the distance is 3e-9 of the norms, d^2 / sum = 1e-17 - below what |a|^2 + |b|^2 - 2 a.b can hold at all, it
    came out as exactly 0 (an adversarial fuzz at the end of round 5).  Such pairs are summed word by word, as the reference does."""
    import math
    import torch
    rng = np.random.default_rng(99)
    n, dim = 130, 256
    counts = rng.integers(0, 2_000_000, size=(n, dim)).astype(np.int64)
    counts[:, 0] += 500_000_000
    near = [(2 * t, 2 * t + 1) for t in range(20)]
    for a, b in near:
        counts[b] = counts[a]
        w = rng.choice(np.arange(1, dim), size=2, replace=False)
        counts[b, w[0]] += 1
        counts[b, w[1]] -= 1 if counts[b, w[1]] > 0 else -1
    totals = counts.sum(1)
    got, st = ctx.pairwise(torch.from_numpy(counts.astype(np.int32)).cuda(), torch.from_numpy(totals).cuda(), "Eucl", want_stats=True)
    got = got.cpu().numpy()
    assert np.array_equal(got, got.T)
    for a, b in near:
        na, nb = int(totals[a]), int(totals[b])
        num = sum((int(x) * nb - int(y) * na) ** 2 for x, y in zip(counts[a], counts[b]))
        want = math.sqrt(num) / (na * nb)
        assert 0 < want < 1e-8 and abs(got[a, b] - want) <= 1e-6 * want, (a, b, got[a, b], want)   # (the quotients c / n are rounded: 1e-16 / 1e-9)


@pytest.mark.parametrize("dim", [4096, 16384])
def test_eucl_three_planes_large_word_space(ctx, dim):
    """Three digit planes at k = 6 / 7: the middle accumulator group adds three digit products per word, so its worst case -
    two records with every count at 2 097 151 (all three digits 127) - is 3 x 16 129 x 16 384 = 7.9e8 < 2^31.  Against the
    oracle; records with totals beyond 2^32 are outside the exact-integer guarantee (G up to 2^56) but not outside rtol."""
    rng = np.random.default_rng(dim)
    n = 150
    counts = rng.integers(0, 700_000, size=(n, dim), dtype=np.uint32)
    counts[rng.random((n, dim)) < 0.3] = 0
    counts[0] = 2_097_151
    counts[1] = 2_097_151
    counts[1, 1::2] = 1_048_575                # (not a near-duplicate of record 0: a Gram form cannot resolve d^2 below 1e-16 of its terms)
    counts[2] = 0
    counts[3, :] = 0
    counts[3, 7] = 1
    totals = counts.sum(1).astype(np.uint64)
    got, st = ctx.pairwise(counts, totals, "Eucl", want_stats=True)
    assert st["kernel_id"] == I8
    freq = oracle.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
    want = oracle.pairwise_block(freq, "Eucl")
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
    assert np.array_equal(got, got.T) and np.all(np.diag(got) == 0.0)
    general = ctx.pairwise(counts, totals, "Eucl", table_path=False)           # the float64 Gram on the same input
    np.testing.assert_allclose(general, got, rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("dim", [4, 64, 256, 1024, 4096, 16384])
def test_spearman_int8_vs_float64_kernel_and_scipy(ctx, dim):
    from scipy.stats import spearmanr
    rng = np.random.default_rng(dim)
    n = 140
    counts = rng.integers(0, 12, size=(n, dim), dtype=np.uint32)          # many ties
    counts[7] = 3                                                          # a constant record -> NaN
    counts[9] = counts[2]
    totals = counts.sum(1).astype(np.uint64)
    got, st = ctx.pairwise(counts, totals, "SC", want_stats=True)
    ref, st0 = ctx.pairwise(counts, totals, "SC", want_stats=True, table_path=False)
    assert st["kernel_id"] == I8 and st0["kernel_id"] == F64
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12, equal_nan=True)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.isnan(got[7, 0]) and np.isnan(got[0, 7])
    assert got[2, 9] == 0.0 and got[9, 2] == 0.0 and np.array_equal(got, got.T, equal_nan=True)
    freq = counts / totals[:, None].astype(np.float64)
    for i, j in [(0, 1), (3, 100), (50, 139), (2, 9)]:
        want = 1.0 - spearmanr(freq[i], freq[j]).correlation
        assert abs(got[i, j] - want) <= RTOL * abs(want) + ATOL
    gf = ctx.pairwise_freq(freq, "SC")
    np.testing.assert_allclose(gf, got, rtol=1e-12, atol=1e-15, equal_nan=True)


def test_spearman_dimension_beyond_two_digits_uses_float64(ctx):
    rng = np.random.default_rng(1)
    freq = rng.random((12, 16385))
    got, st = ctx.pairwise_freq(freq, "SC", want_stats=True)
    assert st["kernel_id"] == F64
    np.testing.assert_allclose(got, oracle.pairwise_block(freq, "SC"), rtol=RTOL, atol=ATOL)


def test_rank_statistics_histogram_and_comparison_paths_agree(ctx):
    """row_order_kernel ranks integer counts below 8192 through a value histogram and everything else (larger
    counts, float64 input) by all-pairs comparison: same ranks, same ties, so KT and SC must agree bit for bit."""
    rng = np.random.default_rng(77)
    counts = rng.integers(0, 40, size=(130, 256), dtype=np.uint32)
    counts[4, 9] = 8191                                  # still the histogram path
    counts[5, 9] = 8192                                  # this record alone falls back to comparisons
    counts[6] = 20000 + rng.integers(0, 3, size=256)     # many ties among large values
    totals = counts.sum(1).astype(np.uint64)
    freq = ctx.frequencies(counts, totals)
    for metric in ("KT", "SC"):
        from_counts = ctx.pairwise(counts, totals, metric)
        from_freq = ctx.pairwise_freq(freq, metric)      # float64 input: comparison path for every record
        assert np.array_equal(from_counts, from_freq, equal_nan=True), metric


def test_frequency_input_of_long_contigs_is_recovered_by_continued_fractions(ctx):
    """Round 4: a long contig has no rare word - its smallest count is in the hundreds or thousands - and the round-1 recovery
    (total = m / smallest frequency for m <= 255) gave up on it, which sent the WHOLE matrix to the float64 kernels (Eucl 12.1
    instead of 5.1 ms on the ragged assembly).  The total now comes out of the continued fraction of a frequency (exact integer
    Euclid on the mantissa), verified bit for bit as before: records with smallest counts 300 .. 40 000 and totals up to 3e7,
    a record whose counts all share a factor with the total (comes back in lowest terms), a record with one word, an empty one
    and short records next to them - the exact int8 kernels run (two and three digit planes) and give what the counts give."""
    rng = np.random.default_rng(11)
    n, dim = 200, 256
    counts = rng.integers(0, 60, size=(n, dim), dtype=np.uint32)                 # short records: smallest count 0 / 1
    counts[0:40] = rng.integers(300, 2000, size=(40, dim))                           # ~200 kb contigs
    counts[40:60] = rng.integers(5_000, 16_000, size=(20, dim))                      # ~1 Mb
    counts[60:70] = rng.integers(40_000, 120_000, size=(10, dim))                    # ~10 Mb: three digit planes
    counts[70] = 6 * rng.integers(50, 500, size=dim)                                 # every count a multiple of 6 (and so is the total)
    counts[71] = 0
    counts[72] = 0
    counts[72, 17] = 123_457
    counts[73] = counts[5]                                                           # a duplicate of a long record
    totals = counts.sum(1).astype(np.uint64)
    assert totals.max() > 2e7
    freq = ctx.frequencies(counts, totals)
    for metric, kid in (("Eucl", I8), ("SC", I8), ("KT", 8)):
        want, st_c = ctx.pairwise(counts, totals, metric, want_stats=True)
        got, st_f = ctx.pairwise_freq(freq, metric, want_stats=True)
        assert st_f["kernel_id"] == st_c["kernel_id"] == kid, (metric, st_f["kernel_id"])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15, equal_nan=True)
    got = ctx.pairwise_freq(freq, "Eucl")
    assert got[5, 73] == 0.0 and got[73, 5] == 0.0
    general = ctx.pairwise_freq(freq, "Eucl", table_path=False)
    np.testing.assert_allclose(got, general, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(got, oracle.pairwise_block(freq, "Eucl"), rtol=RTOL, atol=ATOL)


def test_frequency_input_recovers_the_integer_profiles(ctx):
    """po_pairwise_freq (the reference's own argument type): frequencies that are count / total are traced back to
    the integers - verified bit for bit on the device - and take the same exact kernels as the count entry point;
    anything else (one perturbed value is enough) runs the general float64 kernels."""
    rng = np.random.default_rng(5)
    counts = rng.integers(0, 60, size=(300, 256), dtype=np.uint32)
    counts[7] = 0                                          # empty record
    counts[8, 3:] = 0; counts[8, :3] = (2, 4, 6)           # smallest count 2: n/2 = 6 -> an equivalent reduced pair (c/2, n/2)
    counts[9, :] = 3                                       # constant record; smallest count 3 divides the total
    for i in range(129, 256):                              # records 128 .. 255: permutations of one profile = ONE block with a common
        counts[i] = rng.permutation(counts[128])           # total, so that the equal-total kernels take part (ids 6 / 7 below)
    totals = counts.sum(1).astype(np.uint64)
    freq = ctx.frequencies(counts, totals)
    ids = {"Eucl": 4, "JSD": 6, "BC": 7, "SC": 4, "KT": 8}
    for metric in ("Eucl", "JSD", "BC", "SC", "KT"):
        want, st_c = ctx.pairwise(counts, totals, metric, want_stats=True)
        got, st_f = ctx.pairwise_freq(freq, metric, want_stats=True)
        assert st_f["kernel_id"] == st_c["kernel_id"] == ids[metric], metric
        if metric in ("Eucl", "JSD", "BC"):
            # record 8 is recovered as (1,2,3)/6 instead of (2,4,6)/12, and so is any record whose counts share a
            # factor with its total: same frequencies bit for bit, same mathematics, last-bit differences at most
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15, equal_nan=True)
        else:
            assert np.array_equal(got, want, equal_nan=True)
    bumped = freq.copy()
    bumped[100, 5] = np.nextafter(bumped[100, 5], 1.0)     # no longer count / total
    for metric, general in (("Eucl", 3), ("JSD", 1), ("BC", 2)):
        got, st = ctx.pairwise_freq(bumped, metric, want_stats=True)
        assert st["kernel_id"] == general
        np.testing.assert_allclose(got, ctx.pairwise_freq(freq, metric), rtol=1e-9, atol=1e-12)


def test_eucl_tiles_dealt_to_the_plane_kernels_by_class(ctx):
    """Round 5: from 8 192 records on the host reads the largest count of every 128-record block and deals the tiles to the one- /
    two- / three-plane kernels by class (device-built tile lists, exact grids).  9 000 records: most blocks with counts <= 60, four
    blocks with a record of counts to 5 000, one block with a record of counts to 1 000 000 - so all three classes are present, in
    the triangular matrix, in a row range and in rectangular blocks with mirrors.  Every entry written (the buffers start as NaN),
    exactly symmetric, duplicates exactly 0, float32 = the rounded float64 values, the float64 Gram kernel and the oracle agree."""
    import torch
    rng = np.random.default_rng(2025)
    n, dim = 9000, 256
    counts = rng.integers(0, 60, size=(n, dim), dtype=np.uint32)
    for r in (700, 2100, 2101, 5000, 8990):
        counts[r] = rng.integers(0, 5000, size=dim)
    counts[3333] = rng.integers(0, 1_000_000, size=dim)
    counts[4000] = counts[10]                                   # duplicates across classes of blocks
    counts[8991] = counts[700]
    counts[20] = 0
    totals = counts.sum(1).astype(np.uint64)
    dc, dt = torch.from_numpy(counts.view(np.int32)).cuda(), torch.from_numpy(totals.view(np.int64)).cuda()
    out = torch.full((n, n), float("nan"), dtype=torch.float64, device="cuda")
    _, st = ctx.pairwise(dc, dt, "Eucl", out=out, want_stats=True)
    assert st["kernel_id"] == I8 and st["tiles"] == 71 * 72 // 2
    assert not bool(torch.isnan(out).any())
    assert bool(torch.equal(out, out.T)) and bool((torch.diagonal(out) == 0).all())
    assert float(out[10, 4000]) == 0.0 and float(out[8991, 700]) == 0.0
    out32 = torch.full((n, n), float("nan"), dtype=torch.float32, device="cuda")
    ctx.pairwise(dc, dt, "Eucl", out=out32, dtype="float32")
    assert bool(torch.equal(out32, out.to(torch.float32)))
    general = torch.empty((n, n), dtype=torch.float64, device="cuda")
    ctx.pairwise(dc, dt, "Eucl", out=general, table_path=False)
    assert bool(torch.allclose(out, general, rtol=1e-9, atol=1e-13))
    rows = [0, 20, 700, 2100, 3333, 4000, 8999]
    freq = oracle.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
    want = oracle.pairwise_rows(freq, "Eucl", rows)
    np.testing.assert_allclose(out[rows].cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    # a row range (a rectangular block without a mirror) and two rectangular blocks with mirrors, float64 and float32
    sub = torch.full((1500, n), float("nan"), dtype=torch.float64, device="cuda")
    ctx.pairwise(dc, dt, "Eucl", out=sub, row_begin=2048, row_end=3548)
    assert bool(torch.equal(sub, out[2048:3548]))
    for dtype, full in (("float64", out), ("float32", out32)):
        td = torch.float64 if dtype == "float64" else torch.float32
        blocks = []
        for (r0, r1), (c0, c1) in (((0, 2560), (2560, 9000)), ((3200, 3456), (640, 2200))):
            blocks.append({"rows": (r0, r1), "cols": (c0, c1), "out": torch.full((r1 - r0, c1 - c0), float("nan"), dtype=td, device="cuda"),
                           "mirror": torch.full((c1 - c0, r1 - r0), float("nan"), dtype=td, device="cuda")})
        ctx.pairwise_blocks(dc, dt, "Eucl", blocks, dtype=dtype)
        for b in blocks:
            (r0, r1), (c0, c1) = b["rows"], b["cols"]
            assert bool(torch.equal(b["out"], full[r0:r1, c0:c1])) and bool(torch.equal(b["mirror"], full[c0:c1, r0:r1]))


@pytest.mark.parametrize("n", [1301, 1302, 1304])
def test_float32_store_widths_for_every_alignment_of_the_rows(ctx, n):
    """Round 5: float32 tiles leave through an LDS scratch as 16-byte stores when the rows of the output start on 16-byte
    boundaries, as 8-byte stores on 8-byte boundaries, as single floats otherwise (odd leading dimension) - and element by element,
    with bounds tests, on the edge tiles.  Contiguous n x n outputs with n = 1 301 / 1 302 / 1 304 take the three widths; a block
    that starts inside a tile (rows 77 .., a view whose first column is not a multiple of 4) takes the guarded path.  Eucl (one
    and two digit planes), SC and BC (packed SAD kernel): float32 == the float64 result rounded once, entry for entry."""
    import torch
    rng = np.random.default_rng(n)
    dim = 256
    base = rng.integers(0, 40, size=dim)
    counts = np.stack([rng.permutation(base) for _ in range(n)]).astype(np.uint32)      # one common total: the SAD kernel takes BC
    totals = counts.sum(1).astype(np.uint64)
    big = counts.copy()
    big[5] = rng.permutation(np.concatenate([[3000], rng.integers(0, 300, size=dim - 1)]))   # two digit planes for Eucl
    cases = (("Eucl", counts, totals), ("Eucl", big, big.sum(1).astype(np.uint64)), ("SC", counts, totals), ("BC", counts, totals))
    for metric, c, t in cases:
        dc, dt = torch.from_numpy(c.view(np.int32)).cuda(), torch.from_numpy(t.view(np.int64)).cuda()
        f64 = torch.empty((n, n), dtype=torch.float64, device="cuda")
        ctx.pairwise(dc, dt, metric, out=f64)
        f32 = torch.full((n, n), float("nan"), dtype=torch.float32, device="cuda")
        ctx.pairwise(dc, dt, metric, out=f32, dtype="float32")
        assert bool(torch.equal(f32, f64.to(torch.float32))), (metric, n)
        wide = torch.full((n - 77, n + 3), float("nan"), dtype=torch.float32, device="cuda")   # rows 77 .., written at column 3
        ctx.pairwise(dc, dt, metric, out=wide[:, 3:], dtype="float32", row_begin=77, row_end=n)
        assert bool(torch.equal(wide[:, 3:], f64[77:].to(torch.float32))) and bool(torch.isnan(wide[:, :3]).all()), (metric, n)

"""Reverse-complement folding (csrc/po_fold.hip): `-s both` with a palindromic pattern gives
count[w] == count[rc(w)]; JSD and BC then run over one word per orbit.  The folded result must agree with
the unfolded one and with the oracle, the property must be detected from the data alone, and anything
that lacks it must take the ordinary path."""
import numpy as np
import pytest

from oracle import phyloligo_oracle as oracle
from phyloligo_amd import synthetic

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-12
# folded against unfolded: same mathematics, different summation order.  At D = 4096 the unfolded
# sequential sum of 4096 terms of magnitude ~|ln f| carries ~1e-12 of rounding on its own (measured 8e-13).


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def contigs_ragged(n, seed, lo=300, hi=3000):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        length = int(rng.integers(lo, hi))
        p = [(.2, .3, .3, .2), (.3, .2, .2, .3), (.25, .25, .25, .25), (.35, .15, .15, .35)][i % 4]
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=length, p=p)].copy()
        if i % 7 == 0:
            s[length // 2] = ord("N")
        out.append(s.tobytes())
    out[5] = out[2]                      # a duplicate
    out[9] = b"ACG"                      # shorter than the window: junction words only under `both`
    out[11] = b"NNNNNNNN"                # empty profile
    return out


def pack(contigs):
    seq = np.frombuffer(b"".join(contigs), dtype=np.uint8)
    off = np.zeros(len(contigs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(c) for c in contigs])
    return seq, off


def rc_index(dim):
    k = int(round(np.log(dim) / np.log(4)))
    w = np.arange(dim)
    r = np.zeros(dim, dtype=np.int64)
    x = w.copy()
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 1)
        x >>= 2
    return r


@pytest.mark.parametrize("pattern", ["1", "11", "111", "1111", "11111", "11011", "101", "11011011"])
@pytest.mark.parametrize("metric", ["JSD", "BC"])
def test_both_strands_palindromic_pattern_folds(ctx, pattern, metric):
    contigs = contigs_ragged(150, 77)
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, pattern, "both")
    assert np.array_equal(counts, counts[:, rc_index(counts.shape[1])])        # the symmetry the fold relies on
    folded, st = ctx.pairwise(counts, totals, metric, want_stats=True)
    plain, st0 = ctx.pairwise(counts, totals, metric, want_stats=True, rc_fold=False)
    # (k = 1, and k = 2 for Bray-Curtis: the padded folded layout would be WIDER than the 4 / 16 words it stands for - not
    #  folded, round 5; JSD at k = 2 folds into the same 16 words)
    k_ = pattern.count("1")
    folds = k_ >= 3 or (k_ == 2 and metric == "JSD")
    assert bool(st["rc_folded"]) == folds and not st0["rc_folded"]
    np.testing.assert_allclose(folded, plain, rtol=1e-9, atol=2e-11, equal_nan=True)
    assert np.array_equal(np.isnan(folded), np.isnan(plain))
    freq = oracle.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
    want = oracle.pairwise_distances(freq, metric)
    np.testing.assert_allclose(folded, want, rtol=RTOL, atol=ATOL, equal_nan=True)
    assert np.all(np.diag(folded) == 0.0)
    if metric == "BC":
        assert folded[2, 5] == 0.0 and folded[5, 2] == 0.0
    else:
        assert abs(folded[2, 5]) < 2e-13          # rounding of 2 x 2080 sequential float64 accumulations at D = 4096
    # general kernels only (no table / SAD kernel), rows subset, float32 output, frequency input
    general = ctx.pairwise(counts, totals, metric, table_path=False)
    np.testing.assert_allclose(general, plain, rtol=1e-9, atol=2e-11, equal_nan=True)
    sub = ctx.pairwise(counts, totals, metric, row_begin=17, row_end=93)
    np.testing.assert_allclose(sub, plain[17:93], rtol=1e-9, atol=2e-11, equal_nan=True)
    f32 = ctx.pairwise(counts, totals, metric, dtype="float32")
    np.testing.assert_allclose(f32, plain.astype(np.float32), rtol=1e-6, atol=1e-7, equal_nan=True)
    gf, stf = ctx.pairwise_freq(ctx.frequencies(counts, totals), metric, want_stats=True)
    assert bool(stf["rc_folded"]) == folds
    np.testing.assert_allclose(gf, plain, rtol=1e-9, atol=2e-11, equal_nan=True)


@pytest.mark.parametrize("metric", ["JSD", "BC"])
def test_equal_totals_table_kernels_fold(ctx, metric):
    """fixed-length contigs: the integer-sum table kernel (JSD) and the packed SAD kernel (BC) on folded operands"""
    n = 400
    seq, off = synthetic.contig_bytes(n, 2000, seed=4242)
    for pattern in ("1111", "11011011"):
        counts, totals = ctx.count_profiles(seq, off, pattern, "both")
        folded, st = ctx.pairwise(counts, totals, metric, want_stats=True)
        plain, st0 = ctx.pairwise(counts, totals, metric, want_stats=True, rc_fold=False)
        assert st["rc_folded"] and st["kernel_id"] in (6, 7, 9) and st0["kernel_id"] in (6, 7, 9)
        np.testing.assert_allclose(folded, plain, rtol=1e-9, atol=2e-11)
        freq = counts / totals[:, None].astype(np.float64)
        rows = [0, 1, 57, 399]
        want = oracle.pairwise_rows(freq, metric, rows)
        np.testing.assert_allclose(folded[rows], want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("strand,pattern", [("plus", "1111"), ("minus", "1111"), ("both", "1101"), ("both", "10011")])
def test_asymmetric_profiles_are_not_folded(ctx, strand, pattern):
    contigs = contigs_ragged(100, 5)
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, pattern, strand)
    for metric in ("JSD", "BC"):
        got, st = ctx.pairwise(counts, totals, metric, want_stats=True)
        assert not st["rc_folded"]
        ref = ctx.pairwise(counts, totals, metric, rc_fold=False)
        assert np.array_equal(got, ref, equal_nan=True)                  # the very same kernels ran


def test_one_asymmetric_record_disables_folding(ctx):
    contigs = contigs_ragged(130, 11)
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, "1111", "both")
    counts = counts.copy()
    counts[129, 7] += 1                                                     # rc(7) != 7
    totals = totals.copy()
    totals[129] += 1
    got, st = ctx.pairwise(counts, totals, "JSD", want_stats=True)
    assert not st["rc_folded"]
    freq = oracle.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
    np.testing.assert_allclose(got, oracle.pairwise_distances(freq, "JSD"), rtol=RTOL, atol=ATOL)


def test_dimension_that_is_not_a_power_of_four(ctx):
    rng = np.random.default_rng(3)
    freq = rng.random((70, 50))
    freq /= freq.sum(1, keepdims=True)
    for metric in ("JSD", "BC"):
        got, st = ctx.pairwise_freq(freq, metric, want_stats=True)
        assert not st["rc_folded"]
        np.testing.assert_allclose(got, oracle.pairwise_distances(freq, metric), rtol=RTOL, atol=ATOL)


def test_other_metrics_ignore_the_fold(ctx):
    contigs = contigs_ragged(90, 21)
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, "1111", "both")
    for metric in ("Eucl", "SC"):
        _, st = ctx.pairwise(counts, totals, metric, want_stats=True)
        assert not st["rc_folded"]


def test_long_rows_fold_without_lds_staging(ctx):
    """k = 8: a 65 536-word record does not fit the fold kernel's LDS staging (rc_fold_long_kernel)"""
    contigs = contigs_ragged(24, 9, lo=2000, hi=6000)
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, "11111111", "both")
    assert counts.shape[1] == 65536
    for metric in ("JSD", "BC"):
        folded, st = ctx.pairwise(counts, totals, metric, want_stats=True)
        plain = ctx.pairwise(counts, totals, metric, rc_fold=False)
        assert st["rc_folded"]
        np.testing.assert_allclose(folded, plain, rtol=1e-9, atol=2e-11, equal_nan=True)
        freq = oracle.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
        np.testing.assert_allclose(folded, oracle.pairwise_block(freq, metric), rtol=RTOL, atol=ATOL, equal_nan=True)


@pytest.mark.parametrize("pattern,expect_fold", [("1111", True), ("111", True), ("1", True), ("11", True), ("1001", True),
                                                 ("101", True), ("11011", True), ("1101", False)])
def test_kendall_folds_exactly(ctx, pattern, expect_fold):
    """Kendall's S is an integer: the weighted sum over orbit representatives must reproduce the full sum bit for bit.
    Every palindromic pattern folds (the materialised pair-sign operand orders its word pairs by weight class, whatever
    the number of self-paired words); a pattern that does not read the same in both directions does not."""
    contigs = contigs_ragged(140, 31, lo=200, hi=1500)
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, pattern, "both")
    got, st = ctx.pairwise(counts, totals, "KT", want_stats=True)
    plain, st0 = ctx.pairwise(counts, totals, "KT", want_stats=True, rc_fold=False)
    assert st["rc_folded"] == expect_fold and not st0["rc_folded"]
    assert np.array_equal(got, plain, equal_nan=True)
    valu = ctx.pairwise(counts, totals, "KT", table_path=False)
    assert np.array_equal(got, valu, equal_nan=True)
    sub = ctx.pairwise(counts, totals, "KT", row_begin=3, row_end=77)
    assert np.array_equal(sub, got[3:77], equal_nan=True)
    gf, stf = ctx.pairwise_freq(ctx.frequencies(counts, totals), "KT", want_stats=True)
    assert stf["rc_folded"] == expect_fold and np.array_equal(gf, got, equal_nan=True)
    if counts.shape[1] <= 64:
        from scipy.stats import kendalltau
        freq = ctx.frequencies(counts, totals)
        for i, j in [(0, 1), (2, 5), (20, 99)]:
            tau = kendalltau(freq[i], freq[j], variant="b").statistic
            want = 0.0 if np.isnan(tau) else tau
            assert abs(got[i, j] - want) < 1e-12


def test_kendall_single_strand_not_folded(ctx):
    contigs = contigs_ragged(60, 8)
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, "1111", "plus")
    got, st = ctx.pairwise(counts, totals, "KT", want_stats=True)
    assert not st["rc_folded"]
    assert np.array_equal(got, ctx.pairwise(counts, totals, "KT", table_path=False), equal_nan=True)


@pytest.mark.parametrize("pattern,strand,expect_fold", [("11111", "both", True), ("11111", "plus", False),
                                                        ("11011011", "both", True), ("110011", "minus", False),
                                                        ("1111111", "both", True), ("101111", "both", False)])
def test_kendall_panel_kernel_large_word_spaces(ctx, pattern, strand, expect_fold):
    """k = 5..7 (D = 1 024 .. 16 384): the panelised int8-MFMA kernel (uint16 ranks, 64-word panels, folded or not)
    against the O(D^2) VALU kernel - Kendall's S is an integer, so bit for bit."""
    k = pattern.count("1")
    n = 150 if k <= 6 else 70
    contigs = contigs_ragged(n, 40 + k, lo=400, hi=2500)
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, pattern, strand)
    got, st = ctx.pairwise(counts, totals, "KT", want_stats=True)
    assert st["kernel_id"] == 8 and st["rc_folded"] == expect_fold
    valu, st0 = ctx.pairwise(counts, totals, "KT", want_stats=True, table_path=False)
    assert st0["kernel_id"] == 5
    assert np.array_equal(got, valu, equal_nan=True)
    unfolded = ctx.pairwise(counts, totals, "KT", rc_fold=False)
    assert np.array_equal(got, unfolded, equal_nan=True)
    sub = ctx.pairwise(counts, totals, "KT", row_begin=5, row_end=133 if n > 133 else n - 3)
    assert np.array_equal(sub, got[5:133 if n > 133 else n - 3], equal_nan=True)
    assert np.array_equal(got, got.T, equal_nan=True) and np.all(np.diag(got)[totals > 0] == 1.0)


@pytest.mark.parametrize("k", [1, 2])
def test_word_spaces_whose_folded_layout_would_be_wider_are_not_folded(ctx, k):
    """Round 5, found by a size sweep (tools/exp/f32_regime_sweep.py): at k = 1, 2 the two zero-padded regions of the folded layout
    (Bray-Curtis: 32 + 8 words for the 6 + 4 orbits of k = 2) are WIDER than the 4 / 16 words they stand for, and every workspace of
    the call is sized for the input's width - the operands built from the "folded" matrix ran past their buffers: silently on small
    inputs, a memory fault at 8 191 records (since round 1).  Such word spaces are no longer folded.  8 191 strand-symmetric
    records of 2 kb (counts beyond 255 at k = 2: the general Bray-Curtis kernel, which is the one that faulted), all five metrics,
    against the oracle on sampled rows and against rc_fold=False on the whole matrix."""
    n = 8191
    seq, off = synthetic.contig_bytes(n, 2000, seed=77)
    counts, totals = ctx.count_profiles(seq, off, "1" * k, "both")
    freq = oracle.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
    rows = [0, 1, 4095, 8190]
    for metric in ("BC", "JSD", "KT", "Eucl", "SC"):
        got, st = ctx.pairwise(counts, totals, metric, want_stats=True)
        assert bool(st["rc_folded"]) == (metric == "KT" or (metric == "JSD" and k == 2)), metric   # Kendall folds through its source table
        plain = ctx.pairwise(counts, totals, metric, rc_fold=False)
        if st["rc_folded"] and metric == "JSD":                # (folded float64 sums: another order of summation)
            np.testing.assert_allclose(got, plain, rtol=1e-9, atol=2e-11, equal_nan=True)
        else:
            assert np.array_equal(got, plain, equal_nan=True), metric
        want = oracle.pairwise_rows(freq, metric, rows)
        np.testing.assert_allclose(got[rows], want, rtol=RTOL, atol=1e-9 if metric in ("SC", "KT") else ATOL, equal_nan=True, err_msg=metric)

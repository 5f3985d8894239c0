"""Pin the CPU oracle against vectors produced by the reference's own functions
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import phyloligo_oracle as po

PATTERNS = ["1", "11", "1111", "11111", "1101", "10011", "110101", "11011011"]
STRANDS = ["both", "plus", "minus"]


@pytest.fixture(scope="module")
def prof(golden_dir):
    return np.load(os.path.join(golden_dir, "profiles.npz"))


@pytest.fixture(scope="module")
def dist(golden_dir):
    return np.load(os.path.join(golden_dir, "distances.npz"))


@pytest.mark.parametrize("pattern", PATTERNS)
@pytest.mark.parametrize("strand", STRANDS)
def test_counts_bit_exact(prof, pattern, strand):
    contigs = [bytes(c) for c in prof["contigs"]]
    counts, totals = po.compute_counts(contigs, pattern, strand)
    assert np.array_equal(counts, prof["counts_%s_%s" % (pattern, strand)])
    assert np.array_equal(totals, prof["totals_%s_%s" % (pattern, strand)])
    freq = po.counts_to_frequencies(counts, totals)
    assert np.array_equal(freq, prof["freq_%s_%s" % (pattern, strand)])  # bit-exact float64


@pytest.mark.parametrize("pattern,strand", [("1111", "both"), ("1101", "minus"), ("10011", "plus"), ("11", "both")])
def test_per_window_restatement_equals_the_vectorised_counter_and_the_reference(prof, pattern, strand):
    """oracle.profile_counts_per_window walks the windows one Python string at a time, as the reference does (the cost model of
    bench.py's cpu_baseline stage-1 figure); same integers as the numpy counter and as the reference's own vectors."""
    contigs = [bytes(c) for c in prof["contigs"]][:12] + [b"acgtNNacgtRYacg", b"", b"ACG", b"NNNN"]
    want_c, want_t = po.compute_counts(contigs, pattern, strand)
    for i, c in enumerate(contigs):
        got_c, got_t = po.profile_counts_per_window(c, pattern, strand)
        assert got_t == want_t[i] and np.array_equal(got_c, want_c[i])
    assert np.array_equal(want_c[:12], prof["counts_%s_%s" % (pattern, strand)][:12])


def test_known_answers(prof):
    # facts verified against the imported reference at survey time
    c, t = po.profile_counts("ACGT", "1111", "plus")
    assert t == 1 and c[135] == 1
    assert po.profile_counts("ACG", "1111", "both")[1] == 3
    assert po.profile_counts("ACGTACGTACGTACGTACGTACGT", "1111", "plus")[1] == 21
    assert po.profile_counts("NNNNNNNN", "1111", "both")[1] == 0
    c, t = po.profile_counts("acgtn", "1111", "both")
    assert t == 2 and c[135] == 2
    assert po.profile_counts("ACNGT", "101", "plus")[1] == 0
    c, t = po.profile_counts("ACGTC", "101", "plus")
    assert t == 3  # AG, CT, GC
    idx = lambda w: sum("CGAT".index(ch) * 4 ** (len(w) - 1 - i) for i, ch in enumerate(w))
    assert c[idx("AG")] == 1 and c[idx("CT")] == 1 and c[idx("GC")] == 1


@pytest.mark.parametrize("key", ["1111_both", "11_plus", "1101_minus", "11011011_both"])
@pytest.mark.parametrize("metric", ["Eucl", "JSD", "BC"])
def test_distance_matrices(dist, key, metric):
    pattern, strand = key.split("_")
    contigs = [bytes(c) for c in dist["contigs"]]
    freq = po.compute_frequencies(contigs, pattern, strand)
    assert np.array_equal(freq, dist["freq_" + key])
    want = dist["%s_%s" % (metric, key)]
    if pattern == "11011011":      # D=4096: per-pair python loop is slow, check the block form + 6 rows
        got = po.pairwise_block(freq, metric)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15, equal_nan=True)
        rows = [0, 3, 44, 45, 46, 47]
        np.testing.assert_array_equal(po.pairwise_rows(freq, metric, rows), want[rows])
    else:
        got = po.pairwise_distances(freq, metric)
        np.testing.assert_array_equal(got, want)  # same numpy ops in the same order: bit-exact
        np.testing.assert_allclose(po.pairwise_block(freq, metric), want, rtol=1e-12, atol=1e-15, equal_nan=True)


def test_per_pair_functions(dist):
    freq = dist["freq_1111_both"][:8]
    for name, fn in (("Eucl", po.Eucl), ("JSD", po.JSD), ("KL", po.KL)):
        got = np.array([[fn(a, b) for b in freq] for a in freq])
        np.testing.assert_array_equal(got, dist["pair_%s_1111_both" % name])


@pytest.mark.parametrize("key", ["1111_both", "11_plus", "1101_minus", "11011011_both"])
def test_kt_sc_scipy_pinned(dist, key):
    freq = dist["freq_" + key]
    kt_want, sc_want = dist["scipy_KT_" + key], dist["scipy_SC_" + key]
    n = kt_want.shape[0]
    kt = np.array([[po.KT(freq[i], freq[j]) for j in range(n)] for i in range(n)])
    sc = np.array([[po.SC(freq[i], freq[j]) for j in range(n)] for i in range(n)])
    # SciPy returns NaN for a constant row where the C Clustering Library returns distance 1 (KT 0)
    mask = ~np.isnan(kt_want)
    np.testing.assert_allclose(kt[mask], kt_want[mask], rtol=1e-12, atol=1e-15)
    assert np.all(kt[~mask] == 0.0)
    np.testing.assert_allclose(sc, sc_want, rtol=1e-10, atol=1e-13, equal_nan=True)


def test_mat_bytes(dist):
    m = dist["JSD_1111_both"]
    assert po.mat_bytes(m) == dist["matbytes_JSD_1111_both"].tobytes()
    assert po.mat_bytes(np.array([[0.0, 0.3230493106853566387]])) == b"0.000000000000000000e+00\t3.230493106853566387e-01\n"


def test_c1_synthetic(golden_dir):
    g = np.load(os.path.join(golden_dir, "c1_synthetic.npz"))
    seqs = po.synthetic_contigs(1000, 2000, seed=1001)
    freq = po.compute_frequencies(seqs, "1111", "both")
    assert np.array_equal(freq[:16], g["freq_first16"])
    assert np.array_equal(freq.sum(axis=0), g["freq_colsum"])
    assert np.array_equal(freq.sum(axis=1), g["freq_rowsum"])
    for metric in ("Eucl", "JSD", "BC"):
        np.testing.assert_array_equal(po.pairwise_distances(freq[:16], metric), g["corner_" + metric])
        far = np.array([[po._PAIR[metric](freq[i], freq[j]) for j in (500, 777, 999)] for i in (0, 1, 2)])
        np.testing.assert_array_equal(far, g["far_" + metric])


def test_strand_identities():
    # both = plus + minus + junction words; holds for any pattern
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b"ACGTNacgtR", dtype=np.uint8)
    for pattern in ("1111", "1101", "11011011", "1"):
        W = len(pattern)
        for _ in range(20):
            s = alpha[rng.integers(0, 10, size=int(rng.integers(0, 60)))].tobytes()
            cb, tb = po.profile_counts(s, pattern, "both")
            cp, tp = po.profile_counts(s, pattern, "plus")
            cm, tm = po.profile_counts(s, pattern, "minus")
            tail = s[len(s) - (W - 1):] if W > 1 and len(s) >= W - 1 else (s if W > 1 else b"")
            cj, tj = po.count_pattern(po.select_strand(po.encode(tail), "both"), pattern)
            if len(s) >= W - 1:
                assert np.array_equal(cb, cp + cm + cj) and tb == tp + tm + tj


def test_fasta_parser():
    data = b"\n>a desc\nACGT\nAC GT\r\n\n>b\n>c\nNN\nacgt  \n"
    titles, seqs = po.parse_fasta(data)
    assert titles == ["a desc", "b", "c"]
    assert seqs == [b"ACGTACGT", b"", b"NNacgt"]
    with pytest.raises(ValueError):
        po.parse_fasta(b"ACGT\n>a\nAC\n")


def test_fasta_cases_downstream_of_the_parser(golden_dir):
    """tests/golden/fasta_cases.npz: the reference's own counting code on the records of every hand-built FASTA case
    (IUPAC codes, U, gaps, digits, lower case); the oracle must give the same integers and the same float64 quotients."""
    from tests.fasta_cases import CASES, PROFILE_KEYS
    g = np.load(os.path.join(golden_dir, "fasta_cases.npz"))
    for name, data in CASES.items():
        assert g["data_" + name].tobytes() == data
        titles, seqs = po.parse_fasta(data)
        assert [t.encode("latin-1") for t in titles] == [bytes(t) for t in g["titles_" + name]]
        assert [len(x) for x in seqs] == list(g["seqlens_" + name])
        for pat, strand in PROFILE_KEYS:
            counts, totals = po.compute_counts(seqs, pat, strand)
            assert np.array_equal(counts, g["counts_%s_%s_%s" % (name, pat, strand)]), (name, pat, strand)
            assert np.array_equal(totals, g["totals_%s_%s_%s" % (name, pat, strand)])
            if len(seqs):
                assert np.array_equal(po.counts_to_frequencies(counts, totals), g["freq_%s_%s_%s" % (name, pat, strand)])


WIDE_PATTERNS = ["1" + "0" * 31 + "1", "1101" + "0" * 36 + "1", "11" + "0" * 30 + "101" + "0" * 27 + "11"]


@pytest.mark.parametrize("pattern", WIDE_PATTERNS)
@pytest.mark.parametrize("strand", STRANDS)
def test_wide_patterns_bit_exact(golden_dir, pattern, strand):
    """spaced seeds of 33 / 41 / 64 positions: the reference's own counts (profiles_wide.npz)"""
    g = np.load(os.path.join(golden_dir, "profiles_wide.npz"))
    contigs = [bytes(c) for c in g["contigs"]]
    counts, totals = po.compute_counts(contigs, pattern, strand)
    assert np.array_equal(counts, g["counts_%s_%s" % (pattern, strand)])
    assert np.array_equal(totals, g["totals_%s_%s" % (pattern, strand)])
    assert totals.sum() > 0

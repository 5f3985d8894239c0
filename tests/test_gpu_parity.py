"""GPU parity: the HIP path (through the C ABI) against the golden vectors produced by the
reference's own functions and against the CPU oracle on seeded inputs.  Needs an MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PATTERNS = ["1", "11", "1111", "11111", "1101", "10011", "110101", "11011011"]
STRANDS = ["both", "plus", "minus"]
RTOL = 1e-6      # north_star: float distances within 1e-6 relative of the float64 reference
ATOL = 1e-12     # floor for entries whose reference value is exactly 0 (duplicates, diagonal)


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def prof(golden_dir):
    return np.load(os.path.join(golden_dir, "profiles.npz"))


@pytest.fixture(scope="module")
def dist(golden_dir):
    return np.load(os.path.join(golden_dir, "distances.npz"))


def pack(contigs):
    seq = np.frombuffer(b"".join(contigs), dtype=np.uint8)
    offsets = np.zeros(len(contigs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(c) for c in contigs])
    return seq, offsets


@pytest.mark.parametrize("pattern", PATTERNS)
@pytest.mark.parametrize("strand", STRANDS)
def test_counts_bit_exact_vs_reference(ctx, prof, pattern, strand):
    contigs = [bytes(c) for c in prof["contigs"]]
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, pattern, strand)
    assert np.array_equal(counts.astype(np.int64), prof["counts_%s_%s" % (pattern, strand)])
    assert np.array_equal(totals.astype(np.int64), prof["totals_%s_%s" % (pattern, strand)])
    freq = ctx.frequencies(counts, totals)
    assert np.array_equal(freq, prof["freq_%s_%s" % (pattern, strand)])      # bit-exact float64


WIDE_PATTERNS = ["1" + "0" * 31 + "1", "1101" + "0" * 36 + "1", "11" + "0" * 30 + "101" + "0" * 27 + "11"]


@pytest.mark.parametrize("pattern", WIDE_PATTERNS)
@pytest.mark.parametrize("strand", STRANDS)
def test_wide_window_counts_bit_exact_vs_reference(ctx, golden_dir, pattern, strand):
    """spaced seeds of 33 / 41 / 64 positions (128-bit rolling window): the reference takes any pattern length
    (bin/phyloligo.py:622-628); vectors produced by its cut_sequence_and_count_pattern"""
    g = np.load(os.path.join(golden_dir, "profiles_wide.npz"))
    contigs = [bytes(c) for c in g["contigs"]]
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, pattern, strand)
    assert np.array_equal(counts.astype(np.int64), g["counts_%s_%s" % (pattern, strand)])
    assert np.array_equal(totals.astype(np.int64), g["totals_%s_%s" % (pattern, strand)])


@pytest.mark.parametrize("key", ["1111_both", "11_plus", "1101_minus", "11011011_both"])
@pytest.mark.parametrize("metric", ["Eucl", "JSD", "BC"])
def test_distances_vs_reference(ctx, dist, key, metric):
    pattern, strand = key.split("_")
    contigs = [bytes(c) for c in dist["contigs"]]
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, pattern, strand)
    want = dist["%s_%s" % (metric, key)]
    got = ctx.pairwise(counts, totals, metric)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL, equal_nan=True)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    # frequency-input entry (the reference's own argument) and the no-symmetry path agree
    got_f = ctx.pairwise_freq(dist["freq_" + key], metric)
    np.testing.assert_allclose(got_f, want, rtol=RTOL, atol=ATOL, equal_nan=True)
    got_r = ctx.pairwise(counts, totals, metric, symmetric=False)
    np.testing.assert_allclose(got_r, want, rtol=RTOL, atol=ATOL, equal_nan=True)
    # exact zeros where the reference has exact zeros (diagonal, duplicate record 44 == 3)
    assert np.all(np.diag(got) == 0.0)
    if metric in ("Eucl", "BC"):
        assert got[3, 44] == 0.0 and got[44, 3] == 0.0


@pytest.mark.parametrize("key", ["1111_both", "11_plus", "1101_minus", "11011011_both"])
def test_kt_sc_vs_scipy(ctx, dist, key):
    pattern, strand = key.split("_")
    contigs = [bytes(c) for c in dist["contigs"]]
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, pattern, strand)
    kt_want, sc_want = dist["scipy_KT_" + key], dist["scipy_SC_" + key]
    n = kt_want.shape[0]
    kt = ctx.pairwise(counts, totals, "KT")[:n, :n]
    sc = ctx.pairwise(counts, totals, "SC")[:n, :n]
    mask = ~np.isnan(kt_want)
    np.testing.assert_allclose(kt[mask], kt_want[mask], rtol=RTOL, atol=ATOL)
    assert np.all(kt[~mask] == 0.0)        # constant record: Bio.Cluster distance 1 -> KT 0 (SciPy: NaN)
    np.testing.assert_allclose(sc, sc_want, rtol=RTOL, atol=1e-9, equal_nan=True)


def _random_assembly(n, seed, lo=200, hi=3000, alphabet=b"ACGT"):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    return [alpha[rng.integers(0, len(alpha), size=int(rng.integers(lo, hi)))].tobytes() for _ in range(n)]


@pytest.mark.parametrize("metric", ["Eucl", "JSD", "BC"])
def test_vs_oracle_n300(ctx, metric):
    from oracle import phyloligo_oracle as po
    contigs = _random_assembly(300, 300) + [b"", b"NNNN", b"ACGTNNacgtRYacgtacgtac"]
    contigs.append(contigs[5])
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, "1111", "both")
    oc, ot = po.compute_counts(contigs, "1111", "both")
    assert np.array_equal(counts.astype(np.int64), oc) and np.array_equal(totals.astype(np.int64), ot)
    freq = po.counts_to_frequencies(oc, ot)
    want = po.pairwise_block(freq, metric)
    got = ctx.pairwise(counts, totals, metric)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL, equal_nan=True)
    # independent row blocks (the multi-GPU shard shape) reproduce the full matrix bit for bit
    a = ctx.pairwise(counts, totals, metric, row_begin=0, row_end=130)
    b = ctx.pairwise(counts, totals, metric, row_begin=130, row_end=len(contigs))
    full_rect = ctx.pairwise(counts, totals, metric, symmetric=False)
    assert np.array_equal(np.vstack([a, b]), full_rect, equal_nan=True)
    np.testing.assert_allclose(full_rect, got, rtol=1e-13, atol=1e-15, equal_nan=True)
    # float32 container (memmap variant): one rounding of the float64 value
    got32 = ctx.pairwise(counts, totals, metric, dtype="float32")
    assert got32.dtype == np.float32
    np.testing.assert_array_equal(got32, got.astype(np.float32))


def test_long_and_ragged_records(ctx):
    """Records longer than one chunk (multi-workgroup counting), empty ones, very short ones."""
    from oracle import phyloligo_oracle as po
    rng = np.random.default_rng(9)
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    lens = [0, 1, 3, 4079, 4080, 4081, 4096, 8160, 8161, 30000, 7, 0, 123457, 2, 5000]
    contigs = [alpha[rng.choice(5, size=n, p=[.24, .25, .25, .25, .01])].tobytes() for n in lens]
    seq, offsets = pack(contigs)
    for pattern, strand in [("1111", "both"), ("11011011", "both"), ("10000000000000000000000000000001", "both"),
                            ("1111", "minus"), ("110101", "plus"), ("11111111", "both")]:
        counts, totals = ctx.count_profiles(seq, offsets, pattern, strand)
        oc, ot = po.compute_counts(contigs, pattern, strand)
        assert np.array_equal(counts.astype(np.int64), oc), (pattern, strand)
        assert np.array_equal(totals.astype(np.int64), ot), (pattern, strand)


def test_c1_config_full(ctx, golden_dir):
    """BASELINE config 1: 1000 x 2 kb, k=4, both strands, Eucl -- corners pinned by the reference."""
    from phyloligo_amd import synthetic
    g = np.load(os.path.join(golden_dir, "c1_synthetic.npz"))
    seq, offsets = synthetic.contig_bytes(1000, 2000, seed=1001)
    counts, totals = ctx.count_profiles(seq, offsets, 4, "both")
    assert np.all(totals == 3997)
    freq = ctx.frequencies(counts, totals)
    assert np.array_equal(freq[:16], g["freq_first16"])
    for metric in ("Eucl", "JSD", "BC"):
        m = ctx.pairwise(counts, totals, metric)
        np.testing.assert_allclose(m[:16, :16], g["corner_" + metric], rtol=RTOL, atol=ATOL)
        far = m[np.ix_([0, 1, 2], [500, 777, 999])]
        np.testing.assert_allclose(far, g["far_" + metric], rtol=RTOL, atol=ATOL)
        assert np.array_equal(m, m.T)


def test_jsd_equal_total_table_path_and_mixed_tiles(ctx):
    """Equal-total record blocks take the integer-sum table kernel, the others the general kernel;
    a mixed assembly exercises both inside one matrix, plus tiles that straddle the two."""
    from oracle import phyloligo_oracle as po
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    fixed = [alpha[rng.integers(0, 4, size=1500)].tobytes() for _ in range(300)]          # totals all 2997
    ragged = _random_assembly(200, 12, lo=500, hi=2500)
    contigs = fixed[:256] + ragged[:100] + fixed[256:] + ragged[100:] + [b""]             # blocks: eq, eq, mixed...
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, "1111", "both")
    oc, ot = po.compute_counts(contigs, "1111", "both")
    assert np.array_equal(counts.astype(np.int64), oc)
    want = po.pairwise_block(po.counts_to_frequencies(oc, ot), "JSD")
    got = ctx.pairwise(counts, totals, "JSD")
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ctx.pairwise(counts, totals, "JSD", symmetric=False), want, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ctx.pairwise(counts, totals, "JSD", row_begin=100, row_end=401), want[100:401],
                               rtol=RTOL, atol=ATOL)
    # all-equal assembly (the table kernel only) against the frequency entry point (general kernel only)
    seq2, off2 = pack(fixed)
    c2, t2 = ctx.count_profiles(seq2, off2, "1111", "both")
    lut = ctx.pairwise(c2, t2, "JSD")
    gen = ctx.pairwise_freq(ctx.frequencies(c2, t2), "JSD", table_path=False)
    np.testing.assert_allclose(lut, gen, rtol=1e-9, atol=1e-13)
    assert np.array_equal(ctx.pairwise_freq(ctx.frequencies(c2, t2), "JSD"), lut)      # count2freq output: traced back to the counts
    gen2, st = ctx.pairwise(c2, t2, "JSD", table_path=False, want_stats=True)
    assert st["kernel_id"] == 1 and np.array_equal(gen2, gen)
    o2c, o2t = po.compute_counts(fixed, "1111", "both")
    np.testing.assert_allclose(lut, po.pairwise_block(po.counts_to_frequencies(o2c, o2t), "JSD"), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("metric", ["JSD", "Eucl", "BC", "KT", "SC"])
def test_tournament_blocks_virtual_ranks(ctx, metric, world):
    """The multi-GPU work lists, run rank after rank on one GPU through po_pairwise_blocks_dev: the
    assembled slabs (after the mirror placement that complete_rows does over RCCL) equal the
    single-GPU matrix bit for bit."""
    import torch
    from phyloligo_amd.dist import RowBlockPlan, assemble_virtual
    n = 700 if metric != "KT" else 300
    contigs = _random_assembly(n - 3, 77 + world, lo=300, hi=1500) + [b"", b"ACGTNNNNACGTAC", b"ACGTACGTAC"]
    if metric == "JSD":      # a stretch of equal-total records so that both JSD kernels take part
        contigs[128:384] = _random_assembly(256, 5, lo=900, hi=901)
    seq, offsets = pack(contigs)
    dseq = torch.from_numpy(seq).cuda()
    doff = torch.from_numpy(offsets.astype(np.int64)).cuda()
    pattern = "1111" if metric != "KT" else "111"
    counts, totals = ctx.count_profiles(dseq, doff, pattern, "both")
    full = ctx.pairwise(counts, totals, metric)
    plan = RowBlockPlan(n, world)
    slabs, mirrors = [], []
    for g in range(world):
        slab, mir = plan.allocate(g, counts.device, torch.float64)
        slab.fill_(float("nan"))
        plan.compute(ctx, counts, totals, metric, g, slab, mir)
        slabs.append(slab)
        mirrors.append(mir)
    got = assemble_virtual(plan, slabs, mirrors)
    torch.cuda.synchronize()
    assert torch.equal(torch.isnan(got), torch.isnan(full))
    assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(full, nan=-1.0))


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["eucl_three_planes", "jsd_wide_table"])
def test_tournament_blocks_virtual_ranks_round4_kernels(ctx, case, world):
    """The same check for the two kernel variants of round 4, from synthetic counts: Eucl with counts far beyond two int8 digits
    (three planes) and JSD on equal-total records with counts of 128 .. 255 (the table's wide layout, next to a stretch of
    mixed totals for the general kernel) - the rank-by-rank work lists assemble to the single-GPU matrix bit for bit."""
    import torch
    from phyloligo_amd.dist import RowBlockPlan, assemble_virtual
    rng = np.random.default_rng(31 + world)
    n, dim = 650, 256
    if case == "eucl_three_planes":
        metric, kid = "Eucl", 4
        counts = rng.integers(0, 60, size=(n, dim)).astype(np.uint32)
        counts[40:80] = rng.integers(20_000, 900_000, size=(40, dim))
        counts[300] = 0
        counts[301] = counts[41]
    else:
        metric, kid = "JSD", 6
        base = rng.multinomial(16_000, np.full(dim, 1.0 / dim), size=n).astype(np.uint32)       # totals 16 000, counts ~62 +- 8
        base[:, 0] += 150                                                                           # one word at ~200: wide layout
        counts = base
        counts[:, 2] = 0
        counts[:, 2] = (16_150 + 50 - counts.sum(1)).astype(np.uint32)                             # equal totals again (16 200)
        counts[384:500] = rng.integers(0, 90, size=(116, dim))                                      # mixed totals: general kernel
    totals = counts.sum(1).astype(np.uint64)
    if case == "jsd_wide_table":
        assert len(set(totals[:384].tolist())) == 1 and 127 < counts.max() <= 255
    dc, dt = torch.from_numpy(counts.view(np.int32)).cuda(), torch.from_numpy(totals.view(np.int64)).cuda()
    full, st = ctx.pairwise(dc, dt, metric, want_stats=True)
    assert st["kernel_id"] == kid
    plan = RowBlockPlan(n, world)
    slabs, mirrors = [], []
    for g in range(world):
        slab, mir = plan.allocate(g, dc.device, torch.float64)
        slab.fill_(float("nan"))
        plan.compute(ctx, dc, dt, metric, g, slab, mir)
        slabs.append(slab)
        mirrors.append(mir)
    got = assemble_virtual(plan, slabs, mirrors)
    torch.cuda.synchronize()
    assert torch.equal(torch.isnan(got), torch.isnan(full))
    assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(full, nan=-1.0))


def test_eucl_int8_and_float64_mfma_paths(ctx):
    """Profiles <= 127 take the exact int8-MFMA kernel, larger counts the float64-MFMA kernel; both against
    the oracle, plus the forced general path on the small-count data."""
    from oracle import phyloligo_oracle as po
    small = _random_assembly(333, 21, lo=200, hi=3000) + [b"", b"ACGTACGTAC"]
    small.append(small[7])
    big = _random_assembly(40, 22, lo=30000, hi=60000) + small[:50]          # counts far above 127
    for contigs in (small, big):
        seq, offsets = pack(contigs)
        counts, totals = ctx.count_profiles(seq, offsets, "1111", "both")
        oc, ot = po.compute_counts(contigs, "1111", "both")
        want = po.pairwise_block(po.counts_to_frequencies(oc, ot), "Eucl")
        got, st = ctx.pairwise(counts, totals, "Eucl", want_stats=True)
        assert st["kernel_id"] == 4
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
        assert np.array_equal(got, got.T) and np.all(np.diag(got) == 0.0)
        gen = ctx.pairwise(counts, totals, "Eucl", table_path=False)
        np.testing.assert_allclose(gen, want, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(ctx.pairwise(counts, totals, "Eucl", symmetric=False), want, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(ctx.pairwise(counts, totals, "Eucl", row_begin=11, row_end=77), want[11:77],
                                   rtol=RTOL, atol=ATOL)
    assert counts.max() > 127
    seq, offsets = pack(small)
    counts, totals = ctx.count_profiles(seq, offsets, "1111", "both")
    assert counts.max() <= 127
    got = ctx.pairwise(counts, totals, "Eucl")
    assert got[7, len(small) - 1] == 0.0                     # duplicate records: exactly 0, like (a-b)^2
    got32 = ctx.pairwise(counts, totals, "Eucl", dtype="float32")
    np.testing.assert_array_equal(got32, got.astype(np.float32))


@pytest.mark.parametrize("pattern", ["1111", "11011011"])
def test_bc_equal_total_sad_path_and_mixed_tiles(ctx, pattern):
    """Equal-total record blocks take the packed-byte SAD kernel, the others the float64 kernel."""
    from oracle import phyloligo_oracle as po
    rng = np.random.default_rng(31)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    fixed = [alpha[rng.integers(0, 4, size=1200)].tobytes() for _ in range(300)]
    ragged = _random_assembly(150, 32, lo=400, hi=2000)
    contigs = fixed[:256] + ragged[:70] + fixed[256:] + ragged[70:] + [b"", b""]
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, pattern, "both")
    oc, ot = po.compute_counts(contigs, pattern, "both")
    want = po.pairwise_block(po.counts_to_frequencies(oc, ot), "BC")
    got, st = ctx.pairwise(counts, totals, "BC", want_stats=True)
    assert st["kernel_id"] == 7
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL, equal_nan=True)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    gen, st2 = ctx.pairwise(counts, totals, "BC", want_stats=True, table_path=False)
    assert st2["kernel_id"] == 2
    np.testing.assert_allclose(gen, want, rtol=RTOL, atol=ATOL, equal_nan=True)
    np.testing.assert_allclose(ctx.pairwise(counts, totals, "BC", row_begin=130, row_end=390), want[130:390],
                               rtol=RTOL, atol=ATOL, equal_nan=True)


def test_jsd_general_kernel_near_duplicates(ctx):
    """Records that differ by one base, with different totals (general kernel): tiny JSD values keep their
    relative accuracy, and exact duplicates come out at rounding level."""
    from oracle import phyloligo_oracle as po
    rng = np.random.default_rng(41)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = alpha[rng.integers(0, 4, size=3000)].copy()
    contigs = []
    for i in range(40):
        s = base[:2000 + 13 * i].copy()
        if i % 3 == 1:
            s[500 + i] = alpha[(np.searchsorted(alpha, s[500 + i]) + 1) % 4]     # one substitution
        contigs.append(s.tobytes())
    contigs += [contigs[5], contigs[17]]                                           # exact duplicates
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, "1111", "both")
    oc, ot = po.compute_counts(contigs, "1111", "both")
    want = po.pairwise_block(po.counts_to_frequencies(oc, ot), "JSD")
    got, st = ctx.pairwise(counts, totals, "JSD", want_stats=True)
    off = ~np.eye(len(contigs), dtype=bool)
    assert want[off & (want > 0)].min() < 1e-4                                     # the regime this test is about
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-14)
    assert abs(got[5, 40]) < 5e-14 and abs(got[17, 41]) < 5e-14      # accumulated rounding only


def test_kt_mfma_and_valu_kernels_agree(ctx):
    """Kendall tau through the int8-MFMA pair-sign kernel (default for D <= 256) and through the O(D^2) VALU
    kernel (forced): identical integers S, so identical float64 results; both against the oracle."""
    from oracle import phyloligo_oracle as po
    contigs = _random_assembly(150, 61, lo=200, hi=4000) + [b"", b"ACGTACGTAC", b"A" * 50]
    seq, offsets = pack(contigs)
    for pattern in ("1111", "111", "11", "1"):
        counts, totals = ctx.count_profiles(seq, offsets, pattern, "both")
        fast, st = ctx.pairwise(counts, totals, "KT", want_stats=True)
        slow, st2 = ctx.pairwise(counts, totals, "KT", want_stats=True, table_path=False)
        assert st["kernel_id"] == 8 and st2["kernel_id"] == 5
        assert np.array_equal(fast, slow)
        assert np.array_equal(fast, fast.T)
        assert np.array_equal(fast, ctx.pairwise(counts, totals, "KT", pairdot_i8=True))     # FP4 and int8 operands: same integers
        if pattern in ("111", "11"):
            oc, ot = po.compute_counts(contigs, pattern, "both")
            want = po.pairwise_block(po.counts_to_frequencies(oc, ot), "KT")
            np.testing.assert_allclose(fast, want, rtol=RTOL, atol=ATOL)
        np.testing.assert_array_equal(ctx.pairwise(counts, totals, "KT", row_begin=3, row_end=140), fast[3:140])


@pytest.mark.parametrize("strand", ["both", "plus"])
def test_kt_pairdot_several_tiles(ctx, strand):
    """The materialised pair-sign Gram over more than one 256-record tile, folded (both strands: weight classes
    4/2/1 with accumulator doubling) and unfolded (plus strand: all 32 640 word pairs of k = 4), FP4 and int8
    operands, float32 output, row blocks that do not start on a tile edge - all bit-identical to the O(D^2) VALU kernel."""
    contigs = _random_assembly(701, 77, lo=300, hi=3000)
    contigs[13] = contigs[400]                       # duplicates -> tau exactly 1
    contigs[650] = b"ACGT" * 100                     # heavy ties
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, "1111", strand)
    fast, st = ctx.pairwise(counts, totals, "KT", want_stats=True)
    valu = ctx.pairwise(counts, totals, "KT", table_path=False)
    assert st["kernel_id"] == 8 and bool(st["rc_folded"]) == (strand == "both")
    assert np.array_equal(fast, valu)
    assert np.array_equal(fast, ctx.pairwise(counts, totals, "KT", pairdot_i8=True))
    assert fast[13, 400] == 1.0 and np.all(np.diag(fast) == 1.0)
    np.testing.assert_array_equal(ctx.pairwise(counts, totals, "KT", row_begin=130, row_end=517), fast[130:517])
    f32 = ctx.pairwise(counts, totals, "KT", dtype="float32")
    assert np.array_equal(f32, fast.astype(np.float32))

"""Edge cases of the HIP path: tiny and odd sizes, every word-length regime, ragged row ranges, float32
stores, empty inputs -- against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-12


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def pack(contigs):
    seq = np.frombuffer(b"".join(contigs), dtype=np.uint8)
    offsets = np.zeros(len(contigs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(c) for c in contigs])
    return seq, offsets


def assembly(n, seed, lo=50, hi=600, equal=None):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    p = [.248, .248, .248, .248, .008]
    return [alpha[rng.choice(5, size=(equal or int(rng.integers(lo, hi))), p=p)].tobytes() for _ in range(n)]


@pytest.mark.parametrize("n", [1, 2, 3, 17, 127, 128, 129, 255, 257])
@pytest.mark.parametrize("metric", ["Eucl", "JSD", "BC", "SC", "KT"])
def test_sizes_around_tile_edges(ctx, n, metric):
    from oracle import phyloligo_oracle as po
    if metric == "KT" and n > 129:
        pytest.skip("O(D^2) oracle")
    pattern = "111" if metric in ("KT", "SC") else "1111"
    contigs = assembly(n, 100 + n)
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, pattern, "both")
    oc, ot = po.compute_counts(contigs, pattern, "both")
    assert np.array_equal(counts.astype(np.int64), oc)
    want = po.pairwise_block(po.counts_to_frequencies(oc, ot), metric)
    got = ctx.pairwise(counts, totals, metric)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-9 if metric == "SC" else ATOL, equal_nan=True)
    if n > 2:
        lo, hi = 1, n - 1            # odd row offset, odd leading dimension when n is odd
        part = ctx.pairwise(counts, totals, metric, row_begin=lo, row_end=hi)
        np.testing.assert_allclose(part, want[lo:hi], rtol=RTOL, atol=1e-9 if metric == "SC" else ATOL, equal_nan=True)
    got32 = ctx.pairwise(counts, totals, metric, dtype="float32")
    np.testing.assert_array_equal(got32, got.astype(np.float32))


@pytest.mark.parametrize("pattern", ["1", "11", "111", "11111", "111111", "1111111", "11111111", "101", "1001001"])
def test_every_word_length_regime(ctx, pattern):
    """dim = 4 .. 65 536: padded operand rows (dim < 8), LDS histograms, the global-atomic histogram (k = 8)."""
    from oracle import phyloligo_oracle as po
    k = pattern.count("1")
    n = 40 if k >= 7 else 150
    contigs = assembly(n, 7 + k, lo=100, hi=5000) + [b"", b"ACGT"]
    seq, offsets = pack(contigs)
    for strand in ("both", "minus"):
        counts, totals = ctx.count_profiles(seq, offsets, pattern, strand)
        oc, ot = po.compute_counts(contigs, pattern, strand)
        assert np.array_equal(counts.astype(np.int64), oc), (pattern, strand)
        assert np.array_equal(totals.astype(np.int64), ot)
    freq = po.counts_to_frequencies(oc, ot)
    for metric in ("Eucl", "JSD", "BC"):
        want = po.pairwise_block(freq, metric)
        got = ctx.pairwise(counts, totals, metric)
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL, equal_nan=True)
        gen = ctx.pairwise(counts, totals, metric, table_path=False)
        np.testing.assert_allclose(gen, want, rtol=RTOL, atol=ATOL, equal_nan=True)


@pytest.mark.parametrize("metric", ["JSD", "BC", "Eucl"])
def test_equal_total_blocks_with_ragged_edges(ctx, metric):
    """Fast-path tiles at the matrix edge (N not a multiple of 128), row blocks cutting through tiles."""
    from oracle import phyloligo_oracle as po
    contigs = assembly(300, 5, equal=700)
    contigs = [c.replace(b"N", b"A") for c in contigs]            # all totals equal
    seq, offsets = pack(contigs)
    counts, totals = ctx.count_profiles(seq, offsets, "1111", "both")
    assert len(set(totals.tolist())) == 1
    oc, ot = po.compute_counts(contigs, "1111", "both")
    want = po.pairwise_block(po.counts_to_frequencies(oc, ot), metric)
    got, st = ctx.pairwise(counts, totals, metric, want_stats=True)
    assert st["kernel_id"] in (4, 6, 7, 9)      # 9: BC on thermometer planes (few count levels per word)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
    assert np.array_equal(got, got.T)
    for lo, hi in ((0, 1), (5, 133), (127, 300), (299, 300)):
        np.testing.assert_allclose(ctx.pairwise(counts, totals, metric, row_begin=lo, row_end=hi), want[lo:hi],
                                   rtol=RTOL, atol=ATOL)


def test_empty_and_degenerate_inputs(ctx):
    import phyloligo_amd as pa
    counts, totals = ctx.count_profiles(np.zeros(0, np.uint8), np.zeros(1, np.uint64), "1111", "both")
    assert counts.shape == (0, 256) and totals.shape == (0,)
    assert ctx.pairwise(counts, totals, "JSD").shape == (0, 0)
    seq, offsets = pack([b"", b"", b"NNNN"])
    counts, totals = ctx.count_profiles(seq, offsets, "1111", "both")
    assert counts.sum() == 0 and totals.sum() == 0
    for metric, diag, off in (("Eucl", 0.0, 0.0), ("JSD", 0.0, 0.0), ("KT", 0.0, 0.0)):
        m = ctx.pairwise(counts, totals, metric)
        assert np.all(np.diag(m) == diag) and m[0, 1] == off
    bc = ctx.pairwise(counts, totals, "BC")
    assert np.all(np.diag(bc) == 0.0) and np.isnan(bc[0, 1])          # SciPy: 0/0 off the diagonal
    assert np.all(np.isnan(ctx.pairwise(counts, totals, "SC")))
    with pytest.raises(pa.PhyloligoError):
        ctx.pairwise(counts, totals, "JSD", row_begin=2, row_end=1)
    with pytest.raises(pa.PhyloligoError):
        ctx.count_profiles(seq, offsets, "1111", "sideways")
    with pytest.raises(pa.PhyloligoError):
        ctx.count_profiles(seq, offsets, "1" * 9, "both")


def test_long_records_int8_overflow_and_table_limits(ctx):
    """Counts above 63 / 127 / 255 switch every fast path off on the device; results stay right."""
    from oracle import phyloligo_oracle as po
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    for length, expect_max in ((9000, 64), (20000, 128), (45000, 256)):
        contigs = [alpha[rng.integers(0, 4, size=length)].tobytes() for _ in range(140)]
        seq, offsets = pack(contigs)
        counts, totals = ctx.count_profiles(seq, offsets, "1111", "both")
        assert counts.max() >= expect_max
        oc, ot = po.compute_counts(contigs, "1111", "both")
        freq = po.counts_to_frequencies(oc, ot)
        for metric in ("JSD", "Eucl", "BC"):
            np.testing.assert_allclose(ctx.pairwise(counts, totals, metric), po.pairwise_block(freq, metric),
                                       rtol=RTOL, atol=ATOL)


def test_totals_that_disagree_with_counts(ctx):
    """The C ABI takes totals separately.  Totals that are not the row sums of the counts (equal for all records,
    so the equal-total fast paths are candidates) must still give metric(count/total) exactly as the general
    kernels do: the table kernel's identity sum_w f = 1 does not hold and has to be refused on the device."""
    from oracle import phyloligo_oracle as po
    rng = np.random.default_rng(12)
    counts = rng.integers(0, 30, size=(300, 256), dtype=np.uint32)
    totals = np.full(300, 5000, dtype=np.uint64)                   # row sums are ~3700
    freq = counts / totals[:, None].astype(np.float64)
    for metric in ("JSD", "BC", "Eucl"):
        got = ctx.pairwise(counts, totals, metric)
        ref = ctx.pairwise(counts, totals, metric, table_path=False, rc_fold=False)
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-14)
        want = po.pairwise_block(freq, metric)
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


def test_jsd_table_kernel_covers_counts_up_to_127(ctx):
    """5 kb fixed-length contigs at k=4: counts reach ~70, word sums ~140 - inside the 256-entry table."""
    from oracle import phyloligo_oracle as po
    from phyloligo_amd import synthetic
    seq, off = synthetic.contig_bytes(384, 5000, seed=3)
    counts, totals = ctx.count_profiles(seq, off, "1111", "both")
    assert 63 < counts.max() <= 127 and totals.min() == totals.max()
    table = ctx.pairwise(counts, totals, "JSD")
    general = ctx.pairwise(counts, totals, "JSD", table_path=False)
    np.testing.assert_allclose(table, general, rtol=1e-9, atol=1e-13)
    assert not np.array_equal(table, general)                    # two different kernels did run
    freq = po.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
    np.testing.assert_allclose(table, po.pairwise_block(freq, "JSD"), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("length,strand,lo,hi", [(12_000, "both", 128, 255), (16_000, "both", 128, 255), (20_000, "plus", 100, 255),
                                                 (26_000, "both", 256, 100000)])
def test_jsd_table_kernel_wide_layout_counts_up_to_255(ctx, length, strand, lo, hi):
    """Round 4: fixed-length records of 10 .. 20 kb (a genome cut into windows) have counts of 128 .. 255; the table kernel
    takes them with its wide layout (512 entries x 16 copies, chosen on the device from the largest count) instead of handing
    the matrix to the float64-logarithm kernel.  kernel id 6, against the general kernel and the oracle; a largest count beyond
    255 (last case) still goes to the general kernel, and a narrow call after a wide one on the same context gets the narrow
    layout back (the operands and the table are rebuilt per call)."""
    from oracle import phyloligo_oracle as po
    from phyloligo_amd import synthetic
    seq, off = synthetic.contig_bytes(300, length, seed=length)
    counts, totals = ctx.count_profiles(seq, off, "1111", strand)
    assert lo <= counts.max() <= hi and totals.min() == totals.max()
    table, st = ctx.pairwise(counts, totals, "JSD", want_stats=True)
    general = ctx.pairwise(counts, totals, "JSD", table_path=False)
    np.testing.assert_allclose(table, general, rtol=1e-9, atol=1e-13)
    if counts.max() <= 255:
        assert st["kernel_id"] == 6 and not np.array_equal(table, general)        # the table kernel did the work
    else:
        assert np.array_equal(table, general)                                      # nobody but the general kernel
    freq = po.counts_to_frequencies(counts.astype(np.int64), totals.astype(np.int64))
    np.testing.assert_allclose(table, po.pairwise_block(freq, "JSD"), rtol=RTOL, atol=ATOL)
    assert np.array_equal(table, table.T) and np.all(np.diag(table) == 0.0)
    seq2, off2 = synthetic.contig_bytes(300, 2000, seed=1)
    c2, t2 = ctx.count_profiles(seq2, off2, "1111", "both")
    narrow, st2 = ctx.pairwise(c2, t2, "JSD", want_stats=True)
    assert st2["kernel_id"] == 6
    np.testing.assert_allclose(narrow, ctx.pairwise(c2, t2, "JSD", table_path=False), rtol=1e-9, atol=1e-13)


def test_trim_gives_workspaces_back_and_calls_keep_working():
    """po_ctx_trim frees every grown device workspace (the materialised Kendall operand included); the next call
    allocates again and gives the same bits; a call for another metric releases a pair-dot operand above 1 GB by itself."""
    import torch
    import phyloligo_amd as pa
    rng = np.random.default_rng(9)
    counts = rng.integers(0, 30, size=(700, 256)).astype(np.uint32)
    totals = counts.sum(axis=1).astype(np.uint64)
    with pa.Context(0) as ctx:
        free0 = torch.cuda.mem_get_info(0)[0]
        want = {m: ctx.pairwise(counts, totals, m) for m in ("KT", "JSD", "BC", "Eucl", "SC")}
        used = free0 - torch.cuda.mem_get_info(0)[0]
        assert used > 0
        ctx.trim()
        assert torch.cuda.mem_get_info(0)[0] >= free0 - (8 << 20)          # everything but crumbs (table, flags) is back
        for m, w in want.items():
            assert np.array_equal(ctx.pairwise(counts, totals, m), w, equal_nan=True), m
        ctx.trim()
        ctx.trim()                                                          # idempotent


@pytest.mark.parametrize("dim", [1, 2, 3, 5, 6, 7, 50, 63])
def test_jsd_table_kernel_widths_not_multiple_of_four(ctx, dim):
    """jsd_lut_rows_kernel consumes four words per round; with a width that is not a multiple of 4 the
    surplus slots of the last round must read zero-padded words, not the last real word again.  Equal totals and counts
    <= 127 put every tile on the table kernel; the general kernel (table_path=False) and the oracle are the checks."""
    from oracle import phyloligo_oracle as po
    rng = np.random.default_rng(1000 + dim)
    n, total = 300, 96
    counts = rng.multinomial(total, rng.dirichlet(np.ones(dim) * 0.7), size=n).astype(np.uint32)
    counts[7] = counts[3]                                        # a duplicate pair
    totals = np.full(n, total, dtype=np.uint64)
    assert counts.max() <= 127 and np.array_equal(counts.sum(axis=1), totals)
    got, st = ctx.pairwise(counts, totals, "JSD", want_stats=True)
    assert st["kernel_id"] == 6                                  # the table kernel took part
    gen, st2 = ctx.pairwise(counts, totals, "JSD", table_path=False, want_stats=True)
    assert st2["kernel_id"] == 1
    np.testing.assert_allclose(got, gen, rtol=1e-9, atol=1e-13)
    want = po.pairwise_block(po.counts_to_frequencies(counts.astype(np.int64), totals), "JSD")
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-12)
    assert got[3, 7] < 1e-13 and np.array_equal(got, got.T)


@pytest.mark.parametrize("metric,eq_id,gen_id", [("JSD", 6, 1), ("BC", 7, 2)])
def test_equal_total_kernels_skipped_only_when_no_block_can_qualify(ctx, metric, eq_id, gen_id):
    """The fold pass tells the host whether ANY 128-record block could have one common total (every record of such a block
    shares its total with two block mates); only a certain "no" leaves the table / SAD kernels out.  Ragged totals -> the
    general kernel alone; one uniform block hidden in a ragged assembly, a single-record last block, a uniform assembly ->
    the equal-total kernel takes part; results equal the oracle either way, and chance coincidences of two totals change nothing."""
    from oracle import phyloligo_oracle as po
    rng = np.random.default_rng(5)

    def run(contigs, want_id):
        want_id = (want_id,) if isinstance(want_id, int) else want_id
        seq, offsets = pack(contigs)
        counts, totals = ctx.count_profiles(seq, offsets, "1111", "both")
        got, st = ctx.pairwise(counts, totals, metric, want_stats=True)
        assert st["kernel_id"] in want_id, (st["kernel_id"], want_id)
        oc, ot = po.compute_counts(contigs, "1111", "both")
        np.testing.assert_allclose(got, po.pairwise_block(po.counts_to_frequencies(oc, ot), metric), rtol=RTOL, atol=ATOL, equal_nan=True)

    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    lens = rng.permutation(np.arange(400, 400 + 300))                       # 300 records, no two lengths alike
    ragged = [alpha[rng.integers(0, 4, size=int(n))].tobytes() for n in lens]
    run(ragged, gen_id)
    pairs = list(ragged)
    for i in range(0, 300, 10):                                             # a tenth of the records share one total (a length cut-off):
        pairs[i] = alpha[rng.integers(0, 4, size=400)].tobytes()            # coincidences, but no block of 128 with one total
    run(pairs, gen_id)
    hidden = list(ragged)
    hidden[128:256] = [alpha[rng.integers(0, 4, size=700)].tobytes() for _ in range(128)]     # one uniform block inside
    run(hidden, eq_id)
    run(ragged[:257], eq_id)                                                 # the last block is a single record: trivially uniform
    run([alpha[rng.integers(0, 4, size=600)].tobytes() for _ in range(200)], eq_id if metric == "JSD" else 9)   # 9: thermometer planes

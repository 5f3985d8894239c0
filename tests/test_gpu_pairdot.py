"""Bray-Curtis as a matrix-core Gram over thermometer planes (po_pairdot.hip) against the packed-byte SAD kernel
(the same integers, so bit-identical results), the general float64 kernel and the oracle
('braycurtis' at /root/reference/phylopackage/bin/phyloligo.py:381; SciPy: sum|a-b| / sum|a+b|)."""
import numpy as np
import pytest

from oracle import phyloligo_oracle as po

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-12


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def fixed_length(n, length, seed):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [acgt[rng.integers(0, 4, size=length)].tobytes() for _ in range(n)]


def pack(contigs):
    seq = np.frombuffer(b"".join(contigs), dtype=np.uint8).copy()
    off = np.zeros(len(contigs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(c) for c in contigs])
    return seq, off


@pytest.mark.parametrize("pattern,strand,n,length", [("11011011", "both", 700, 2000), ("11011011", "plus", 300, 2000),
                                                     ("11111", "both", 520, 2000), ("110011", "minus", 260, 800)])
def test_bc_thermometer_equals_sad(ctx, pattern, strand, n, length):
    contigs = fixed_length(n, length, seed=len(pattern) * 100 + n)
    contigs[17] = contigs[3]                                   # duplicates: exactly 0
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, pattern, strand)
    fast, st = ctx.pairwise(counts, totals, "BC", want_stats=True)
    sad, st_sad = ctx.pairwise(counts, totals, "BC", want_stats=True, pairdot=False)
    assert st["kernel_id"] == 9 and st_sad["kernel_id"] == 7
    assert bool(st["rc_folded"]) == (strand == "both" and pattern == pattern[::-1])
    assert np.array_equal(fast, sad)                           # same integer numerators, same epilogue
    assert np.array_equal(fast, ctx.pairwise(counts, totals, "BC", pairdot_i8=True))
    assert np.array_equal(fast, fast.T) and np.all(np.diag(fast) == 0.0) and fast[17, 3] == 0.0
    general = ctx.pairwise(counts, totals, "BC", table_path=False)
    np.testing.assert_allclose(fast, general, rtol=1e-12, atol=1e-15)
    oc, ot = po.compute_counts(contigs, pattern, strand)
    assert np.array_equal(counts.astype(np.int64), oc)
    want = po.pairwise_block(po.counts_to_frequencies(oc, ot), "BC")
    np.testing.assert_allclose(fast, want, rtol=RTOL, atol=ATOL)
    # row blocks off the tile grid, float32 output
    np.testing.assert_array_equal(ctx.pairwise(counts, totals, "BC", row_begin=130, row_end=259), fast[130:259])
    assert np.array_equal(ctx.pairwise(counts, totals, "BC", dtype="float32"), fast.astype(np.float32))


def test_bc_thermometer_not_taken_when_totals_differ(ctx):
    """Blocks of 128 records with equal totals inside but different totals between them, a ragged assembly and an
    empty record: the matrix is left to the SAD / general kernels and stays right."""
    a = fixed_length(128, 2000, 1) + fixed_length(128, 2100, 2)
    seq, off = pack(a)
    counts, totals = ctx.count_profiles(seq, off, "11011011", "both")
    got, st = ctx.pairwise(counts, totals, "BC", want_stats=True)
    assert st["kernel_id"] == 7
    oc, ot = po.compute_counts(a, "11011011", "both")
    np.testing.assert_allclose(got, po.pairwise_block(po.counts_to_frequencies(oc, ot), "BC"), rtol=RTOL, atol=ATOL)
    rng = np.random.default_rng(9)
    b = [c[:int(rng.integers(500, 2000))] for c in fixed_length(200, 2000, 3)] + [b""]
    seq, off = pack(b)
    counts, totals = ctx.count_profiles(seq, off, "11011011", "both")
    got, st = ctx.pairwise(counts, totals, "BC", want_stats=True)
    assert st["kernel_id"] in (2, 7)
    oc, ot = po.compute_counts(b, "11011011", "both")
    np.testing.assert_allclose(got, po.pairwise_block(po.counts_to_frequencies(oc, ot), "BC"), rtol=RTOL, atol=ATOL, equal_nan=True)


def test_blockwise_equal_totals_mixed_tiles(ctx):
    """Round-1 bug: 128-record blocks that are uniform inside but differ from each other (reads of two lengths, sorted)
    passed the 'every block qualifies' test, the general kernel was left out and the mixed tiles were never written."""
    a = fixed_length(128, 2000, 11) + fixed_length(128, 1900, 12) + fixed_length(40, 2000, 13)
    seq, off = pack(a)
    for pattern, metric in (("1111", "JSD"), ("1111", "BC"), ("11011011", "BC"), ("111", "JSD")):
        counts, totals = ctx.count_profiles(seq, off, pattern, "both")
        out = np.full((len(a), len(a)), np.nan)
        ctx.pairwise(counts, totals, metric, out=out)
        assert not np.isnan(out).any()
        oc, ot = po.compute_counts(a, pattern, "both")
        np.testing.assert_allclose(out, po.pairwise_block(po.counts_to_frequencies(oc, ot), metric), rtol=RTOL, atol=ATOL)


def test_bc_thermometer_through_blocks(ctx):
    """The block-list entry point (the per-rank work lists of the multi-GPU plan) over the thermometer path."""
    import torch
    from phyloligo_amd.dist import RowBlockPlan, assemble_virtual
    contigs = fixed_length(900, 2000, seed=5)
    seq, off = pack(contigs)
    dseq, doff = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    counts, totals = ctx.count_profiles(dseq, doff, "11011011", "both")
    full, st = ctx.pairwise(counts, totals, "BC", want_stats=True)
    assert st["kernel_id"] == 9
    plan = RowBlockPlan(900, 3)
    slabs, mirrors = [], []
    for g in range(3):
        slab, m = plan.allocate(g, counts.device, torch.float64)
        s2 = plan.compute(ctx, counts, totals, "BC", g, slab, m, want_stats=True)
        assert s2["kernel_id"] == 9
        slabs.append(slab)
        mirrors.append(m)
    assert torch.equal(assemble_virtual(plan, slabs, mirrors), full)


@pytest.mark.parametrize("pattern,strand,n", [("11111", "both", 300), ("11111", "plus", 200), ("110111", "both", 130),
                                              ("11011011", "both", 140)])
def test_kt_pairdot_large_word_spaces(ctx, pattern, strand, n):
    """Kendall above 256 words (k = 5, 6): uint16 ranks kept transposed in HBM, the pair-sign operand materialised
    once (up to 1 MB per record), against the panel kernel (pairdot=False) and the O(D^2) vector kernel - S is an
    integer, so all three agree bit for bit; FP4 and int8 operands too."""
    rng = np.random.default_rng(n)
    contigs = [c[:int(rng.integers(600, 2000))] for c in fixed_length(n, 2000, seed=n + 1)]
    contigs[5] = contigs[9]
    seq, off = pack(contigs)
    counts, totals = ctx.count_profiles(seq, off, pattern, strand)
    fast, st = ctx.pairwise(counts, totals, "KT", want_stats=True)
    panel, st_p = ctx.pairwise(counts, totals, "KT", want_stats=True, pairdot=False)
    assert st["kernel_id"] == 8 and st_p["kernel_id"] == 8
    assert np.array_equal(fast, panel)
    assert np.array_equal(fast, ctx.pairwise(counts, totals, "KT", pairdot_i8=True))
    if counts.shape[1] <= 1024:
        assert np.array_equal(fast, ctx.pairwise(counts, totals, "KT", table_path=False))
    assert np.array_equal(fast, fast.T) and fast[5, 9] == 1.0 and np.all(np.diag(fast) == 1.0)
    np.testing.assert_array_equal(ctx.pairwise(counts, totals, "KT", row_begin=7, row_end=101), fast[7:101])


def test_kendall_k6_operand_beyond_24_gb(ctx):
    """Round 4: the materialised pair-sign operand may take up to 96 GB of the 288 GB of HBM (24 GB before).  Kendall at k = 6 is
    1.05 MB of FP4 signs per record: 25 000 records = 26 GB, which round 3 sent to the panel kernel.  The matrix-core Gram and the
    panel kernel (pairdot=False) agree bit for bit on row blocks at both ends of the matrix; one row against the SciPy-pinned oracle."""
    import torch
    from oracle import phyloligo_oracle as po
    from phyloligo_amd import synthetic
    n = 25_000
    seq, off = synthetic.contig_bytes(n, 2000, seed=606)
    counts, totals = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "111111", "both")
    a, st = ctx.pairwise(counts, totals, "KT", row_begin=0, row_end=256, want_stats=True)
    assert st["kernel_id"] == 8 and st["rc_folded"]
    b, st_p = ctx.pairwise(counts, totals, "KT", row_begin=0, row_end=256, pairdot=False, want_stats=True)
    assert st_p["kernel_id"] == 8                               # the panel kernel reports the same family id
    assert torch.equal(a, b)
    a2 = ctx.pairwise(counts, totals, "KT", row_begin=n - 130, row_end=n)
    b2 = ctx.pairwise(counts, totals, "KT", row_begin=n - 130, row_end=n, pairdot=False)
    assert torch.equal(a2, b2)
    freq = po.counts_to_frequencies(counts[:1].cpu().numpy().astype(np.int64), totals[:1].cpu().numpy())
    cols = np.array([1, 2, 777, n - 1])
    fc = po.counts_to_frequencies(counts[torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.int64), totals[torch.from_numpy(cols).cuda()].cpu().numpy())
    want = np.array([po.KT(freq[0], fc[i]) for i in range(len(cols))])
    np.testing.assert_allclose(a[0, torch.from_numpy(cols).cuda()].cpu().numpy(), want, rtol=1e-6, atol=1e-12)
    ctx.trim()                                                  # 26 GB of operand go back

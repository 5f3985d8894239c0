"""BASELINE.json's full-size configurations (50 000 contigs x 2 kb) on one GPU, checked through properties that do
not need an O(N^2) oracle: exact symmetry, diagonal, value range, agreement of independent code paths
(row-block call vs full matrix, folded vs unfolded operands), and a few complete rows against the oracle."""
import numpy as np
import pytest

from oracle import phyloligo_oracle as oracle
from phyloligo_amd import synthetic

pytestmark = pytest.mark.gpu
N = 50_000


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def profiles(ctx, pattern, seed):
    import torch
    seq, off = synthetic.contig_bytes(N, 2000, seed=seed)
    dseq = torch.from_numpy(seq.copy()).cuda()
    doff = torch.from_numpy(off.astype(np.int64)).cuda()
    return ctx.count_profiles(dseq, doff, pattern, "both")


def is_symmetric(out, block=8192, nan_ok=False):
    """out[i, j] == out[j, i] bit for bit; nan_ok: a nan must face a nan (Spearman of a constant row)"""
    import torch
    n = out.shape[0]
    for i in range(0, n, block):
        for j in range(i, n, block):
            a = out[i:i + block, j:j + block]
            b = out[j:j + block, i:i + block].T
            if nan_ok:
                a, b = torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)
            if not torch.equal(a, b):
                return False
    return True


@pytest.mark.parametrize("config,pattern,metric,seed,lo,hi", [
    ("C2", "1111", "JSD", 50001, 0.0, float(np.log(2.0))),
    ("C3", "1111", "Eucl", 50001, 0.0, float(np.sqrt(2.0))),
    ("C5", "11011011", "BC", 50005, 0.0, 1.0),
])
def test_full_size_matrix_properties(ctx, config, pattern, metric, seed, lo, hi):
    import torch
    counts, totals = profiles(ctx, pattern, seed)
    assert int(totals.min()) == int(totals.max()) == 2 * (2000 - len(pattern) + 1) + len(pattern) - 1   # 3 997 / 3 993
    out = torch.empty((N, N), dtype=torch.float64, device="cuda")
    _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
    assert st["pairs"] == N * N // 2 and st["rc_folded"] == (metric != "Eucl")
    assert is_symmetric(out)
    assert bool((torch.diagonal(out) == 0).all())
    assert float(out.min()) >= lo and float(out.max()) <= hi * (1 + 1e-12)
    assert not bool(torch.isnan(out).any())
    # independent paths: a row block computed without the symmetry shortcut, and the unfolded operands
    sub = ctx.pairwise(counts, totals, metric, row_begin=31_000, row_end=31_300)
    assert torch.equal(sub, out[31_000:31_300]) if metric != "JSD" else torch.allclose(sub, out[31_000:31_300], rtol=1e-12, atol=1e-15)
    plain = ctx.pairwise(counts, totals, metric, row_begin=7, row_end=135, rc_fold=False)
    assert torch.allclose(plain, out[7:135], rtol=1e-9, atol=1e-13)
    # complete rows against the oracle
    freq = oracle.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
    for r in (0, 24_999, N - 1):
        want = oracle.pairwise_block(np.vstack([freq[r:r + 1], freq]), metric, 0, 1)[0, 1:]
        want[r] = 0.0
        np.testing.assert_allclose(out[r].cpu().numpy(), want, rtol=1e-6, atol=1e-12)
    # a checksum of checksums: row sums equal column sums exactly (symmetry) and their total is reproducible
    rs, cs = out.sum(dim=1), out.sum(dim=0)
    assert torch.allclose(rs, cs, rtol=1e-12)
    del out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("metric,diag,lo,hi", [("KT", 1.0, -1.0, 1.0), ("SC", 0.0, 0.0, 2.0)])
def test_full_size_rank_metrics(ctx, metric, diag, lo, hi):
    """Kendall (matrix-core Gram over the materialised pair-sign operand, reverse-complement folded) and Spearman (exact
    int8 Gram on doubled ranks) at the C2 size: symmetry, diagonal, range, row blocks off the tile grid, the unfolded /
    int8 operand variants bit for bit, and two rows against the SciPy-pinned oracle on 3 000 columns each."""
    import torch
    counts, totals = profiles(ctx, "1111", 50001)
    out = torch.empty((N, N), dtype=torch.float64, device="cuda")
    _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
    assert st["pairs"] == N * N // 2
    assert is_symmetric(out)
    assert bool((torch.diagonal(out) == diag).all())
    assert float(out.min()) >= lo - 1e-12 and float(out.max()) <= hi + 1e-12 and not bool(torch.isnan(out).any())
    assert torch.equal(ctx.pairwise(counts, totals, metric, row_begin=40_001, row_end=40_260), out[40_001:40_260])
    if metric == "KT":
        assert st["kernel_id"] == 8 and st["rc_folded"]
        assert torch.equal(ctx.pairwise(counts, totals, "KT", row_begin=130, row_end=390, rc_fold=False), out[130:390])
        assert torch.equal(ctx.pairwise(counts, totals, "KT", row_begin=130, row_end=390, pairdot_i8=True), out[130:390])
    freq = oracle.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
    for r, c0 in ((0, 20_000), (N - 1, 0)):
        want = oracle.pairwise_block(np.vstack([freq[r:r + 1], freq[c0:c0 + 3000]]), metric, 0, 1)[0, 1:]
        np.testing.assert_allclose(out[r, c0:c0 + 3000].cpu().numpy(), want, rtol=1e-6, atol=1e-12)
    del out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("pattern,strand,seed", [("1111", "both", 50001), ("11011011", "both", 50005), ("1111", "plus", 50001),
                                                  ("1101", "both", 50001), ("111111", "minus", 50001), ("1101", "minus", 50001),
                                                  ("1101" + "0" * 36 + "1", "both", 50001)])
def test_full_size_profiles(ctx, pattern, strand, seed):
    """Stage 1 on the whole 50 000-contig assembly (register-string fast path for every strand mode of the short patterns,
    two histograms per wave for the non-palindromic `both`; 128-bit rolling windows on the general path for the 41-wide seed):
    row sums are the totals, the totals are the window counts the reference's arithmetic gives, '-s both' profiles are
    reverse-complement symmetric, and 150 contigs drawn over the whole range equal the oracle bit for bit."""
    import torch
    seq, off = synthetic.contig_bytes(N, 2000, seed=seed)
    counts, totals = ctx.count_profiles(torch.from_numpy(seq.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), pattern, strand)
    W, k = len(pattern), pattern.count("1")
    per_strand = 2000 - W + 1
    want_total = 2 * per_strand + W - 1 if strand == "both" else per_strand
    assert int(totals.min()) == int(totals.max()) == want_total
    assert torch.equal(counts.sum(dim=1, dtype=torch.int64), totals.to(torch.int64))
    if strand == "both":                                        # seq + revcomp(seq) is its own reverse complement: word w <-> rc(w)
        d = torch.arange(4 ** k, device="cuda")
        digits = torch.stack([(d >> (2 * (k - 1 - i))) & 3 for i in range(k)], dim=1)      # C0 G1 A2 T3, first base first
        rc = ((digits.flip(1) ^ 1) << (2 * (k - 1 - torch.arange(k, device="cuda")))).sum(dim=1)
        if pattern == pattern[::-1]:
            assert torch.equal(counts, counts[:, rc])
    rng = np.random.default_rng(seed)
    pick = np.sort(rng.choice(N, size=150, replace=False))
    recs = [seq[int(off[i]):int(off[i + 1])].tobytes() for i in pick]
    oc, ot = oracle.compute_counts(recs, pattern, strand)
    assert np.array_equal(counts[torch.from_numpy(pick).cuda()].cpu().numpy().astype(np.int64), oc)


# ---- a ragged, dirty assembly at full size: the kernels every REAL input gets -------------------------
@pytest.fixture(scope="module")
def ragged(ctx):
    """50 000 contigs, log-normal lengths 1 - 200 kb (0.33 Gb), four base compositions, N runs, lower case, IUPAC codes
    (phyloligo_amd.synthetic.ragged_assembly) + one record that is nothing but N (an empty profile among 49 999 full ones)."""
    import torch
    seq, off = synthetic.ragged_assembly(N, seed=2024)
    seq = seq.copy()
    off64 = off.astype(np.int64)
    seq[off64[12_345]:off64[12_346]] = ord("N")
    counts, totals = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off64).cuda(), "1111", "both")
    return seq, off64, counts, totals


def test_ragged_assembly_profiles(ctx, ragged):
    """Stage 1 on multi-chunk records (up to 200 kb = ~100 chunks each), separators in most of them: 200 sampled records -
    the longest, the all-N one and the first / last among them - equal the oracle's counts bit for bit; row sums are the
    totals; no two blocks of 128 records share a total (every stage-2 tile is a "mixed" tile)."""
    import torch
    seq, off, counts, totals = ragged
    lens = np.diff(off)
    assert lens.min() >= 1000 and lens.max() == 200_000 and 3.0e8 < off[-1] < 3.6e8
    assert torch.equal(counts.sum(dim=1, dtype=torch.int64), totals.to(torch.int64))
    t = totals.cpu().numpy()
    assert t[12_345] == 0 and int(counts[12_345].abs().sum()) == 0
    for b in range(0, N, 128):
        assert np.unique(t[b:b + 128]).size > 1
    assert 127 < int(counts.max()) <= 16_383                       # two 7-bit digit planes for the exact Gram
    rng = np.random.default_rng(4)
    pick = np.unique(np.concatenate([rng.choice(N, size=196, replace=False), [0, N - 1, 12_345, int(np.argmax(lens))]]))
    recs = [seq[off[i]:off[i + 1]].tobytes() for i in pick]
    oc, ot = oracle.compute_counts(recs, "1111", "both")
    assert np.array_equal(counts[torch.from_numpy(pick).cuda()].cpu().numpy().astype(np.int64), oc)
    assert np.array_equal(t[pick], ot)
    # the same records on one strand and under a spaced pattern (general path, two histograms): 40 of them
    sub = pick[::5]
    for pattern, strand in (("1111", "minus"), ("1101", "both"), ("11011011", "plus")):
        c2, t2 = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off).cuda(), pattern, strand)
        oc, ot = oracle.compute_counts([seq[off[i]:off[i + 1]].tobytes() for i in sub], pattern, strand)
        assert np.array_equal(c2[torch.from_numpy(sub).cuda()].cpu().numpy().astype(np.int64), oc), (pattern, strand)
        assert np.array_equal(t2.cpu().numpy()[sub], ot)
        del c2, t2


@pytest.mark.parametrize("metric,kernel_id,diag,cols", [("JSD", 1, 0.0, None), ("BC", 2, 0.0, None), ("Eucl", 4, 0.0, None),
                                                        ("SC", 4, 0.0, 8000), ("KT", 8, 1.0, 4000)])
def test_ragged_assembly_matrix(ctx, ragged, metric, kernel_id, diag, cols):
    """The whole 50 000 x 50 000 matrix of the ragged assembly through the kernels real data gets - valu_tile_kernel<JSD> /
    <BC> for every tile (kernel ids 1 / 2: the fold pass reports that no 128-record block can have one common total, so the
    table / SAD kernels - which would own no tile - are not even prepared; round 3 launched them: ids 6 / 7), the two-plane exact int8 Gram for Eucl and
    SC, the pair-dot Gram for KT - against the oracle: three complete rows (rank metrics: three rows x 8 000 / 4 000 columns;
    the oracle's Kendall is O(D^2) per pair) at rtol 1e-6, exact symmetry, the diagonal, the empty record's row, a row block
    off the tile grid computed without the symmetry shortcut, and for Eucl the forced float64 Gram on one row block."""
    import torch
    seq, off, counts, totals = ragged
    out = torch.empty((N, N), dtype=torch.float64, device="cuda")
    _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
    assert st["kernel_id"] == kernel_id and st["pairs"] == N * N // 2
    assert st["rc_folded"] == (metric in ("JSD", "BC", "KT"))
    assert is_symmetric(out, nan_ok=(metric == "SC"))
    d = torch.diagonal(out)
    if metric == "KT":                                   # the empty record is a constant row: tau 0 with everything, itself included
        assert float(d[12_345]) == 0.0 and bool((torch.cat([d[:12_345], d[12_346:]]) == 1.0).all())
    elif metric == "SC":                                 # a constant row has no rank correlation: nan, as SciPy gives
        assert bool(torch.isnan(out[12_345]).all()) and bool((torch.cat([d[:12_345], d[12_346:]]) == 0.0).all())
    else:
        assert bool((d == diag).all())
    sub = ctx.pairwise(counts, totals, metric, row_begin=33_333, row_end=33_600)
    assert torch.allclose(sub, out[33_333:33_600], rtol=1e-12, atol=1e-15, equal_nan=True)
    freq = oracle.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
    longest = int(np.argmax(np.diff(off)))
    for r in (0, longest, N - 1) + ((12_345,) if cols is None else ()):
        c0 = 0 if cols is None else (r // 2 if r + 1 < N else N - cols)
        c1 = N if cols is None else c0 + cols
        want = oracle.pairwise_block(np.vstack([freq[r:r + 1], freq[c0:c1]]), metric, 0, 1)[0, 1:]
        if c0 <= r < c1:
            want[r - c0] = diag if totals[r] > 0 or metric != "KT" else 0.0
        got = out[r, c0:c1].cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-12, equal_nan=True, err_msg="%s row %d" % (metric, r))
    if metric in ("JSD", "BC"):
        # the empty record against a full one: 1/2 ln 2 (phylodist.py:22-24 masks its terms) and 1 (SciPy braycurtis)
        assert abs(float(out[12_345, 7]) - (0.5 * np.log(2.0) if metric == "JSD" else 1.0)) < 1e-12
    if metric == "Eucl":
        f64 = ctx.pairwise(counts, totals, "Eucl", row_begin=20_000, row_end=20_256, table_path=False)
        assert torch.allclose(f64, out[20_000:20_256], rtol=1e-9, atol=1e-13)
    del out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("pattern", ["111111", "11011011"])
def test_ragged_assembly_large_alphabet(ctx, pattern):
    """The large word space of BASELINE config 5 (D = 4 096) on RAGGED, dirty data - config 5 itself is equal-length contigs, which
    take the thermometer / SAD / table kernels; a real assembly at k = 6 takes the general kernels and two digit planes: 8 000
    contigs of 1 - 60 kb, JSD / BC / Eucl / SC rows against the oracle, symmetry, and three Kendall pairs (the oracle's Kendall is
    O(D^2) = 8 million comparisons per pair)."""
    import torch
    n = 8_000
    seq, off = synthetic.ragged_assembly(n, seed=77, median=3000, sigma=0.9, lo=1000, hi=60_000)
    off64 = off.astype(np.int64)
    counts, totals = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off64).cuda(), pattern, "both")
    pick = [0, 1234, int(np.argmax(np.diff(off64))), n - 1]
    oc, ot = oracle.compute_counts([seq[off64[i]:off64[i + 1]].tobytes() for i in pick], pattern, "both")
    assert np.array_equal(counts[torch.tensor(pick).cuda()].cpu().numpy().astype(np.int64), oc)
    freq = oracle.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    for metric, kid in (("JSD", 1), ("BC", 2), ("Eucl", 4), ("SC", 4)):
        _, st = ctx.pairwise(counts, totals, metric, out=out, want_stats=True)
        assert st["kernel_id"] == kid, (metric, st["kernel_id"])
        assert is_symmetric(out, nan_ok=(metric == "SC"))
        for r in pick[:3]:
            want = oracle.pairwise_block(np.vstack([freq[r:r + 1], freq[:2000]]), metric, 0, 1)[0, 1:]
            if r < 2000:
                want[r] = 0.0
            np.testing.assert_allclose(out[r, :2000].cpu().numpy(), want, rtol=1e-6, atol=1e-12, err_msg="%s row %d" % (metric, r))
    kt, st = ctx.pairwise(counts, totals, "KT", row_begin=0, row_end=2, want_stats=True)
    assert st["kernel_id"] == 8
    for j in (1, 4000, n - 1):
        np.testing.assert_allclose(float(kt[0, j]), oracle.KT(freq[0], freq[j]), rtol=1e-6, atol=1e-12)
    del out
    ctx.trim()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("pattern,strand,big", [("1111", "both", 2_500_000_000), ("11011011", "plus", 2_500_000_000),
                                                ("111", "plus", 4_400_000_000)])
def test_profiles_of_more_than_4_gib_of_sequence(ctx, pattern, strand, big):
    """Maximum sizes of stage 1: byte offsets beyond 2^32 and ONE record longer than 2^31 (2.5 GB) / 2^32 (4.4 GB) bytes - a
    chromosome-scale scaffold; the reference has no limit but memory (str slicing, bin/phyloligo.py:601-631).  Layout: 48 dirty,
    ragged probe records, the long record, 2 000 records of 0.5 - 1.5 MB, the same probe records again (their bytes start
    beyond 4 GiB).  The probe records equal the oracle at both ends bit for bit; every filler record has its window count
    ('both' = the record and its reverse complement as one string: 2 L - W + 1, phyloligo.py:141); row sums are the totals; the long
    record's counts equal a histogram of its words computed by torch in pieces (plus strand, contiguous words)."""
    import torch
    torch.cuda.empty_cache()                                   # what earlier tests left in torch's cache is not "used"
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * (1 << 30):
        pytest.skip("needs ~40 GB of free HBM")
    pseq, poff = synthetic.ragged_assembly(n=48, seed=7, median=20000, sigma=1.0, lo=100, hi=300000)[:2]
    poff = np.asarray(poff, dtype=np.int64)
    n_probe = len(poff) - 1
    rng = np.random.default_rng(11)
    fill_lens = np.concatenate([[big], rng.integers(500_000, 1_500_000, size=2000)]).astype(np.int64)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    total_fill = int(fill_lens.sum())
    p = torch.from_numpy(np.asarray(pseq)).cuda()
    seq = torch.empty(2 * p.numel() + total_fill, dtype=torch.uint8, device="cuda")
    seq[:p.numel()] = p
    seq[-p.numel():] = p
    for a in range(0, total_fill, 1 << 30):
        b = min(total_fill, a + (1 << 30))
        seq[p.numel() + a:p.numel() + b] = lut[torch.randint(0, 4, (b - a,), device="cuda", generator=g, dtype=torch.uint8).long()]
    lens = np.concatenate([np.diff(poff), fill_lens, np.diff(poff)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    assert off[-1] == seq.numel() and off[-1 - n_probe] > (1 << 32)
    counts, totals = ctx.count_profiles(seq, torch.from_numpy(off).cuda(), pattern, strand)
    c = counts.cpu().numpy().view(np.uint32).astype(np.int64)
    t = totals.cpu().numpy().astype(np.int64)
    oc, ot = oracle.compute_counts([np.asarray(pseq)[poff[i]:poff[i + 1]].tobytes() for i in range(n_probe)], pattern, strand)
    assert np.array_equal(c[:n_probe], oc) and np.array_equal(t[:n_probe], ot)
    assert np.array_equal(c[-n_probe:], oc) and np.array_equal(t[-n_probe:], ot)            # the same bytes beyond 4 GiB
    span = len(pattern)
    assert np.array_equal(t[n_probe:-n_probe], (fill_lens * 2 if strand == "both" else fill_lens) - span + 1)
    assert np.array_equal(c.sum(axis=1), t)
    if strand == "plus" and set(pattern) == {"1"}:
        a0, L = int(off[n_probe]), int(fill_lens[0])
        code = torch.zeros(256, dtype=torch.int64, device="cuda")
        for i, ch in enumerate(b"CGAT"):                                 # the reference's column order (count2freq :653-658)
            code[ch] = i
        hist = torch.zeros(4 ** span, dtype=torch.int64, device="cuda")
        for s in range(0, L - span + 1, 1 << 28):
            e = min(L - span + 1, s + (1 << 28))
            w = torch.zeros(e - s, dtype=torch.int64, device="cuda")
            for j in range(span):
                w = w * 4 + code[seq[a0 + s + j:a0 + e + j].long()]
            hist += torch.bincount(w, minlength=4 ** span)
        assert torch.equal(hist.cpu(), torch.from_numpy(c[n_probe]))

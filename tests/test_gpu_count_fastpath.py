"""Stage 1, fast path of count_kernel (forward words of a window of at most 16 positions on chunks that hold nothing
but A/C/G/T: digits by bit operations, windows out of a packed register string, wave-uniform totals and junction
words) - bit-exact against the oracle (pinned by the goldens to the reference's cut_sequence_and_count_pattern /
select_strand, /root/reference/phylopackage/bin/phyloligo.py:124-149, :601-631) on the cases its index arithmetic can
get wrong: record lengths around the 32-base lane, the 2 016-start chunk and the window length, records that start at
every alignment, the chunk whose tail lies before the staged range, the last record of the buffer, lower case, and
clean records next to dirty ones (which took the general path inside the same launch until round 5 and stay on the fast path
since: test_dirty_chunks_on_the_fast_path)."""
import numpy as np
import pytest

from oracle import phyloligo_oracle as oracle

pytestmark = pytest.mark.gpu

# window <= 16, at most 4 runs, at most 7 ones (LDS histogram); the palindromic ones take the fast path under "both" too
PATTERNS = ["1", "11", "111", "1111", "11111", "111111", "1111111", "101", "1001", "11011", "1110111", "11011011",
            "1000000000000001", "1100110011", "110101011", "1101", "10011", "1011101", "1100000000000001", "110100111",
            "1011", "10110111"]
LENGTHS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 17, 30, 31, 32, 33, 34, 63, 64, 65, 100, 1999, 2000, 2001, 2013, 2014, 2015,
           2016, 2017, 2018, 2019, 2020, 2030, 2031, 2032, 2033, 2047, 2048, 2049, 4031, 4032, 4033, 4034, 4040, 6047,
           6048, 6049, 6051, 10000]


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def clean(rng, length, lower=False):
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=length)]
    if lower and length:
        s = s.copy()
        m = rng.random(length) < 0.3
        s[m] |= 0x20
    return s.tobytes()


def check(ctx, records, pattern, strands=("both", "plus")):
    seq = np.frombuffer(b"".join(records), dtype=np.uint8)
    off = np.zeros(len(records) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in records])
    for strand in strands:
        counts, totals = ctx.count_profiles(seq, off, pattern, strand)
        oc, ot = oracle.compute_counts(records, pattern, strand)
        bad = np.nonzero((counts.astype(np.int64) != oc).any(axis=1))[0]
        assert bad.size == 0, (pattern, strand, "records", bad[:8], [len(records[i]) for i in bad[:8]])
        assert np.array_equal(np.asarray(totals).astype(np.int64), np.asarray(ot).astype(np.int64)), (pattern, strand)


@pytest.mark.parametrize("pattern", PATTERNS)
def test_boundary_lengths(ctx, pattern):
    """all three strand modes: `minus` and `both` with a pattern that is not its own mirror image ride on the same
    register string (forward words of the reversed pattern, relabelled when the histogram is written out; round 3)"""
    rng = np.random.default_rng(len(pattern) * 977 + pattern.count("1"))
    records = [clean(rng, n, lower=(i % 3 == 0)) for i, n in enumerate(LENGTHS)]
    order = rng.permutation(len(records))          # starts at every alignment, a different record last in the buffer
    check(ctx, [records[i] for i in order], pattern, strands=("both", "plus", "minus"))


@pytest.mark.parametrize("seed", range(6))
def test_clean_and_dirty_neighbours(ctx, seed):
    rng = np.random.default_rng(4242 + seed)
    pattern = PATTERNS[int(rng.integers(0, len(PATTERNS)))]
    records = []
    for _ in range(60):
        n = int(rng.choice(LENGTHS)) + int(rng.integers(0, 3))
        r = bytearray(clean(rng, n, lower=rng.random() < 0.3))
        if n > 10 and rng.random() < 0.35:          # one separator somewhere: that chunk takes the general path
            r[int(rng.integers(0, n))] = ord("N")
        records.append(bytes(r))
    check(ctx, records, pattern, strands=("both", "plus", "minus"))


@pytest.mark.parametrize("pattern", PATTERNS)
def test_dirty_chunks_on_the_fast_path(ctx, pattern):
    """Round 5: a chunk with bytes that are not A/C/G/T stays on the fast path - the starts whose window touches such a byte
    are cleared from the lanes' validity masks (the touched bits cross lanes with the register string's halo), the words are
    counted instead of computed, and the junction windows of `-s both` go through the per-base code.  Dirt where that can go
    wrong: the first and the last base of a record, inside its last W - 1 bases (the junction), around the 32-base lane, around
    the 2 016-start chunk, runs longer than the window, IUPAC codes and lower-case n, a record that is nothing but dirt."""
    rng = np.random.default_rng(len(pattern) * 31 + pattern.count("1"))
    W = len(pattern)
    records = []
    for n in (1, 2, W - 1, W, W + 1, 2 * W, 31, 32, 33, 64, 65, 100, 2000, 2015, 2016, 2017, 2048, 4031, 4033, 6049, 9000):
        if n < 1:
            continue
        spots = {0, n - 1, n // 2, max(0, n - W), max(0, n - W + 1), max(0, n - 2), min(n - 1, 31), min(n - 1, 32), min(n - 1, 33),
                 min(n - 1, 63), min(n - 1, 64), min(n - 1, 2015), min(n - 1, 2016), min(n - 1, 2017), min(n - 1, 2047), min(n - 1, 2048)}
        for k, spot in enumerate(sorted(min(max(x, 0), n - 1) for x in spots)):
            if k % 3 != int(rng.integers(0, 3)):
                continue
            r = bytearray(clean(rng, n, lower=rng.random() < 0.3))
            r[spot] = ord(str(rng.choice(list("NnRYKMSWBDHVU-*"))))
            if rng.random() < 0.4:                                  # a run that is longer than the window, somewhere else
                a = int(rng.integers(0, n))
                r[a:a + W + 3] = b"N" * len(r[a:a + W + 3])
            records.append(bytes(r))
    records += [b"N" * 40, b"NNNN", bytes(clean(rng, 2500)), b"n" * 2100, b"ACGT" * 3 + b"R" + b"ACGT" * 600]
    order = rng.permutation(len(records))
    check(ctx, [records[i] for i in order], pattern, strands=("both", "plus", "minus"))


def test_long_record_and_windows(ctx):
    """One 300 kb record (149 chunks of one record: atomics into one row), and overlapping byte ranges of it as
    records (the Kount scan's shape: ranges start at arbitrary offsets of the same buffer)."""
    rng = np.random.default_rng(7)
    big = clean(rng, 300000)
    check(ctx, [clean(rng, 777), big, clean(rng, 5)], "1111")
    check(ctx, [big], "11011011", strands=("both",))
    # chunks of one record sharing a histogram in the relabelled layouts; a separator sends one of its chunks down the general path
    dirty = bytearray(big)
    dirty[123456] = ord("N")
    check(ctx, [clean(rng, 3000), bytes(dirty), clean(rng, 2017)], "1101", strands=("both", "minus"))
    check(ctx, [bytes(dirty)[:9000], clean(rng, 4100)], "110100111", strands=("both", "minus"))
    seq = np.frombuffer(big, dtype=np.uint8)
    begins = np.arange(0, 300000 - 5000, 1777, dtype=np.uint64)
    ends = begins + np.uint64(5000)
    for pattern, strand in (("1111", "both"), ("111", "plus"), ("1101", "both"), ("10011", "minus")):
        counts, totals = ctx.count_profiles_ranges(seq, begins, ends, pattern, strand)
        oc, ot = oracle.compute_counts([big[int(b):int(e)] for b, e in zip(begins, ends)], pattern, strand)
        assert np.array_equal(counts.astype(np.int64), oc)
        assert np.array_equal(np.asarray(totals).astype(np.int64), np.asarray(ot).astype(np.int64))


@pytest.mark.parametrize("pattern,strand,seed", [("1111", "both", 0), ("1111", "both", 1), ("1111", "plus", 2), ("111111", "minus", 3),
                                                 ("1101", "both", 4), ("11011011", "both", 5), ("1011", "plus", 6), ("1111111", "both", 7),
                                                 ("110100111", "both", 8), ("1" + "0" * 38 + "11", "both", 9), ("10000000000000000000000000000011", "minus", 10)])
def test_long_records_go_through_segment_rows(ctx, pattern, strand, seed):
    """Records of more than 512 chunks (~1 Mb) add their workgroups' histograms into scratch rows per 128-chunk segment instead
    of one row of the count matrix (po_count.hip, kSegChunks; profiles/r04_stage1.txt).  Who owns a segment is decided by its
    first chunk: two long records back to back, long records that start in the middle of a segment and right behind short ones,
    a record of exactly 512 / 513 chunks, dirt (N runs, lower case, IUPAC) - all against the oracle, bit for bit, and again in a
    second call on the same context (the scratch rows have to be zero again) with the records in reverse order."""
    import torch
    rng = np.random.default_rng(900 + seed)
    span = 2016                                                       # window starts per chunk (kChunkSpan)
    lens = []
    for _ in range(int(rng.integers(5, 60))):
        lens.append(int(rng.integers(1, 5000)))
    lens += [int(rng.integers(1_100_000, 1_400_000)), int(rng.integers(1_050_000, 1_300_000))]      # two long ones back to back
    lens += [int(rng.integers(1, 3000)) for _ in range(int(rng.integers(1, 40)))]
    lens += [512 * span + len(pattern) - 1, 7, 512 * span + len(pattern), 513 * span]                  # 512 and 513 chunks
    lens += [int(rng.integers(2_000_000, 3_000_000))]
    lens += [int(rng.integers(1, 3000)) for _ in range(3)]
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for L in lens:
        r = alphabet[rng.integers(0, 4, size=L)].copy()
        for _ in range(max(1, L // 20000)):                               # dirt: N runs, lower case, an IUPAC code
            a = int(rng.integers(0, L))
            r[a:a + int(rng.integers(1, 60))] = ord("N")
            a = int(rng.integers(0, L))
            r[a:a + int(rng.integers(1, 200))] |= 0x20
        if L > 10:
            r[int(rng.integers(0, L))] = ord("R")
        recs.append(r.tobytes())
    oc, ot = oracle.compute_counts(recs, pattern, strand)

    def run(order):
        seq = np.frombuffer(b"".join(recs[i] for i in order), dtype=np.uint8)
        off = np.concatenate([[0], np.cumsum([len(recs[i]) for i in order])]).astype(np.int64)
        c, t = ctx.count_profiles(torch.from_numpy(seq.copy()).cuda(), torch.from_numpy(off).cuda(), pattern, strand)
        return c.cpu().numpy().view(np.uint32).astype(np.int64), t.cpu().numpy().astype(np.int64)

    order = list(range(len(recs)))
    c, t = run(order)
    assert np.array_equal(t, ot) and np.array_equal(c, oc)
    c, t = run(order[::-1])
    assert np.array_equal(t, ot[::-1]) and np.array_equal(c, oc[::-1])

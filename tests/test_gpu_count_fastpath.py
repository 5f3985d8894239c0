"""Stage 1, fast path of count_kernel (forward words of a window of at most 16 positions on chunks that hold nothing
but A/C/G/T: digits by bit operations, windows out of a packed register string, wave-uniform totals and junction
words) - bit-exact against the oracle (pinned by the goldens to the reference's cut_sequence_and_count_pattern /
select_strand, /root/reference/phylopackage/bin/phyloligo.py:124-149, :601-631) on the cases its index arithmetic can
get wrong: record lengths around the 32-base lane, the 2 016-start chunk and the window length, records that start at
every alignment, the chunk whose tail lies before the staged range, the last record of the buffer, lower case, and
clean records next to dirty ones (which take the general path inside the same launch)."""
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

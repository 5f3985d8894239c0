"""Seeded fuzz of stage 1: random spaced-word patterns (window up to 64, up to 8 ones) on random records with
separators, lower case and IUPAC symbols, all strands - bit-exact against the oracle (which the goldens pin to the
reference's cut_sequence_and_count_pattern / select_strand)."""
import numpy as np
import pytest

from oracle import phyloligo_oracle as oracle

pytestmark = pytest.mark.gpu
ALPHABET = np.frombuffer(b"ACGTacgtNnRYKMSWBDHVU-*", dtype=np.uint8)
WEIGHTS = np.array([20, 20, 20, 20, 4, 4, 4, 4, 2, 1] + [0.25] * 13)
WEIGHTS = WEIGHTS / WEIGHTS.sum()


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def random_pattern(rng, wide=False):
    k = int(rng.integers(1, 9))
    w = int(rng.integers(33, 65)) if wide else int(rng.integers(k, 33))
    pos = np.sort(rng.choice(w, size=k, replace=False))
    if rng.random() < 0.7:                       # usually a '1' at both ends, as real spaced seeds have
        pos[0], pos[-1] = 0, w - 1
        pos = np.unique(pos)
    pat = ["0"] * w
    for p in pos:
        pat[p] = "1"
    return "".join(pat)


def random_records(rng, n):
    out = []
    for _ in range(n):
        length = int(rng.choice([0, 1, 3, 7, 31, 32, 33, 100, 2031, 2032, 2033, 2100, 5000, 20000], p=None))
        length = max(0, length + int(rng.integers(-2, 3)))
        s = ALPHABET[rng.choice(len(ALPHABET), size=length, p=WEIGHTS)]
        if length > 50 and rng.random() < 0.3:   # a run of separators
            a = int(rng.integers(0, length - 20))
            s[a:a + int(rng.integers(1, 20))] = ord("N")
        out.append(s.tobytes())
    return out


@pytest.mark.parametrize("seed", range(12))
def test_random_patterns_and_records(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    pattern = random_pattern(rng)
    records = random_records(rng, 40)
    seq = np.frombuffer(b"".join(records), dtype=np.uint8)
    off = np.zeros(len(records) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in records])
    for strand in ("both", "plus", "minus"):
        counts, totals = ctx.count_profiles(seq, off, pattern, strand)
        oc, ot = oracle.compute_counts(records, pattern, strand)
        assert np.array_equal(counts.astype(np.int64), oc), (pattern, strand)
        assert np.array_equal(totals.astype(np.int64), ot), (pattern, strand)


@pytest.mark.parametrize("seed", range(8))
def test_random_wide_patterns_and_records(ctx, seed):
    """windows of 33..64 positions: the 128-bit rolling registers, long records across chunks included"""
    rng = np.random.default_rng(5000 + seed)
    pattern = random_pattern(rng, wide=True)
    records = random_records(rng, 30)
    seq = np.frombuffer(b"".join(records), dtype=np.uint8)
    off = np.zeros(len(records) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in records])
    for strand in ("both", "plus", "minus"):
        counts, totals = ctx.count_profiles(seq, off, pattern, strand)
        oc, ot = oracle.compute_counts(records, pattern, strand)
        assert np.array_equal(counts.astype(np.int64), oc), (pattern, strand)
        assert np.array_equal(totals.astype(np.int64), ot), (pattern, strand)

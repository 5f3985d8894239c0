"""Synthetic assemblies of BASELINE.md section 4 / SURVEY.md section 8d: i.i.d. uniform ACGT
contigs from numpy.random.default_rng(seed), headers >c{i:07d}, 80-column FASTA wrapping."""
import numpy as np

SEEDS = {"C1": 1001, "C2": 50001, "C3": 50001, "C4": 200001, "C5": 50005}
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def contig_bytes(n, length=2000, seed=1001):
    """uint8[n*length] concatenated sequence bytes and uint64[n+1] offsets.  Contig i is
    rng.integers(0, 4, size=length, dtype=uint8) drawn in order, mapped through b"ACGT"."""
    rng = np.random.default_rng(seed)
    seq = np.empty(n * length, dtype=np.uint8)
    for i in range(n):
        seq[i * length:(i + 1) * length] = _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]
    offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(length)
    return seq, offsets


def fasta_bytes(seq, offsets, width=80):
    out = []
    raw = seq.tobytes()
    for i in range(len(offsets) - 1):
        s = raw[int(offsets[i]):int(offsets[i + 1])]
        out.append(b">c%07d\n" % i)
        for p in range(0, len(s), width):
            out.append(s[p:p + width] + b"\n")
    return b"".join(out)


def contig_bytes_range(n, length, seed, lo, hi):
    """Contigs [lo,hi) of the assembly contig_bytes(n, length, seed) without materialising the rest
    (the generator is advanced past the first `lo` contigs draw by draw)."""
    rng = np.random.default_rng(seed)
    for _ in range(lo):
        rng.integers(0, 4, size=length, dtype=np.uint8)
    m = hi - lo
    seq = np.empty(m * length, dtype=np.uint8)
    for i in range(m):
        seq[i * length:(i + 1) * length] = _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]
    offsets = np.arange(m + 1, dtype=np.uint64) * np.uint64(length)
    return seq, offsets


# ---- a real-assembly-like input -------------------------------------------------------------------------------------
# What PhylOligo is run on (README of the reference: draft assemblies; SURVEY section 7: contig lengths 10^2 .. 10^7): ragged
# lengths, no two word totals alike, counts far above a byte, four "species" of different base composition, and the
# dirt real FASTA carries - runs of N, soft-masked (lower-case) stretches, IUPAC ambiguity codes.  Deterministic in `seed`.
SPECIES = ((.20, .30, .30, .20), (.30, .20, .20, .30), (.25, .25, .25, .25), (.35, .15, .15, .35))    # A, C, G, T
_IUPAC = np.frombuffer(b"RYKMSWBDHVryn", dtype=np.uint8)


def ragged_lengths(n=50000, seed=2024, median=4000, sigma=1.0, lo=1000, hi=200000):
    rng = np.random.default_rng(seed)
    return np.clip(np.exp(rng.normal(np.log(median), sigma, size=n)), lo, hi).astype(np.int64)


def ragged_assembly(n=50000, seed=2024, median=4000, sigma=1.0, lo=1000, hi=200000, dirt=True):
    """(seq uint8[sum L], offsets uint64[n+1]).  Contig i: log-normal length (median 4 kb, clipped to 1 .. 200 kb: ~0.33 Gb
    at n = 50 000), base composition SPECIES[i % 4].  dirt=True adds, per record: N runs (about one per 20 kb, 1 .. 60 long,
    a record in 50 gets one of 200 .. 2 000), lower-case stretches (a record in 5 gets 1 .. 3 of 50 .. 2 000 bases) and single
    IUPAC codes (about one per 30 kb)."""
    lens = ragged_lengths(n, seed, median, sigma, lo, hi)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    total = int(offsets[-1])
    rng = np.random.default_rng(seed + 1)
    seq = np.empty(total, dtype=np.uint8)
    off = offsets.astype(np.int64)
    for sp, probs in enumerate(SPECIES):                       # one draw per species, cut into that species' records in order
        idx = np.arange(sp, n, len(SPECIES))
        size = int(lens[idx].sum())
        cum = np.cumsum(np.asarray(probs, dtype=np.float64))[:3].astype(np.float32)
        bases = _ACGT[np.searchsorted(cum, rng.random(size, dtype=np.float32), side="right").astype(np.uint8)]
        at = 0
        for i in idx:
            seq[off[i]:off[i + 1]] = bases[at:at + lens[i]]
            at += int(lens[i])
        del bases
    if dirt:
        # N runs: start positions over the whole assembly, clipped to their record
        rec_of = lambda pos: np.searchsorted(off, pos, side="right") - 1            # noqa: E731
        k = max(1, total // 20000)
        starts = rng.integers(0, total, size=k)
        runs = rng.integers(1, 61, size=k)
        big = rng.choice(n, size=max(1, n // 50), replace=False)
        starts = np.concatenate([starts, off[big] + (rng.random(big.size) * lens[big]).astype(np.int64)])
        runs = np.concatenate([runs, rng.integers(200, 2001, size=big.size)])
        ends = np.minimum(starts + runs, off[rec_of(starts) + 1])
        for a, b in zip(starts, ends):
            seq[a:b] = ord("N")
        # soft-masked stretches
        soft = rng.choice(n, size=max(1, n // 5), replace=False)
        for i in soft:
            for _ in range(int(rng.integers(1, 4))):
                a = off[i] + int(rng.integers(0, lens[i]))
                b = min(off[i + 1], a + int(rng.integers(50, 2001)))
                seq[a:b] |= 0x20                                   # ASCII lower case ('N' -> 'n' too)
        # single ambiguity codes
        k = max(1, total // 30000)
        seq[rng.integers(0, total, size=k)] = _IUPAC[rng.integers(0, _IUPAC.size, size=k)]
    return seq, offsets

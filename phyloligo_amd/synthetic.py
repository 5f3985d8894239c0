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

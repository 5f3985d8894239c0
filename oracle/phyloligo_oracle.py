"""CPU oracle for the PhylOligo all-by-all contig distance path.

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement (numpy, float64) of the
reference algorithm.  It is imported by `tests/`, by `__graft_entry__.smoke()` and by the
`cpu_baseline` leg of `bench.py` -- as the checker / the timed CPU baseline -- and by
nothing else.  The product (`phyloligo_amd/`) never imports it and has no CPU fallback.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real reference in the
build container (stand-in modules for its absent third-party imports)
and stores the outputs of the reference's own functions; `tests/test_oracle_golden.py`
checks every function below against those vectors (counts bit-exact, Eucl/JSD/BC
bit-exact or <=1e-15).  KT and SC cannot be produced by the reference here (KT needs
Biopython's compiled Bio.Cluster, SC raises NameError in the reference as shipped):
those two are pinned against SciPy (kendalltau variant 'b', spearmanr) and labelled so.

What stays dependent on stand-ins for Biopython (absent here), exactly:
  * parse_fasta below restates SimpleFastaParser from Biopython's published behaviour; no
    reference run can pin it.  Everything DOWNSTREAM of the parser is pinned: the records of
    every hand-built FASTA case (tests/fasta_cases.py) went through the reference's own
    select_strand / cut_sequence_and_count_pattern / compute_frequency
    (tests/golden/fasta_cases.npz, checked by test_fasta_cases_downstream_of_the_parser);
  * Seq.reverse_complement on the 'minus' / 'both' strands: A/C/G/T/N of either case are
    complemented by the stand-in exactly as Biopython does.  IUPAC ambiguity codes
    (RYKMSWBDHV), gaps, digits and any other byte are not A/C/G/T before or after any
    complement table, so they split words identically (stand-in independent).  The ONE
    dependent character is 'U' / 'u': here a separator on both strands; a Biopython release
    whose DNA table maps U -> A would count the minus-strand copy of a U as 'A'.

Reference lines restated (paths relative to /root/reference/phylopackage/):
  bin/phyloligo.py:124-149   select_strand
  bin/phyloligo.py:601-631   cut_sequence_and_count_pattern
  bin/phyloligo.py:633-661   count2freq
  bin/phyloligo.py:663-691   compute_frequency
  bin/phyloligo.py:847-877   compute_frequencies_joblib
  bin/phyloligo.py:364-392   compute_distances_joblib  (sklearn pairwise_distances semantics)
  bin/phyloligo.py:1059-1066 numpy.savetxt of the matrices
  core/phylodist.py:12-85    posdef_check_value, KL, Eucl, JSD, KT, BC, SC
"""
from __future__ import annotations

import io
import numpy as np

# ----------------------------------------------------------------------------------------
# Alphabet.  count2freq (phyloligo.py:653) enumerates itertools.product(("C","G","A","T"))
# so the digit of a base is C=0, G=1, A=2, T=3, first letter most significant.  In this
# coding the Watson-Crick complement is digit XOR 1.
# ----------------------------------------------------------------------------------------
INVALID = 255
_CODE = np.full(256, INVALID, dtype=np.uint8)
for _d, _ch in enumerate(b"CGAT"):
    _CODE[_ch] = _d
    _CODE[_ch + 32] = _d  # lower case: the reference upper-cases after strand selection (:683)

METRICS = ("Eucl", "JSD", "KT", "BC", "SC")
STRANDS = ("both", "plus", "minus")


def _as_bytes(seq) -> bytes:
    if isinstance(seq, str):
        return seq.encode("latin-1", errors="replace")
    return bytes(seq)


def encode(seq) -> np.ndarray:
    """bytes/str -> uint8 digit per base (255 for anything that is not A/C/G/T, any case)."""
    return _CODE[np.frombuffer(_as_bytes(seq), dtype=np.uint8)]


def select_strand(codes: np.ndarray, strand: str = "both") -> np.ndarray:
    """phyloligo.py:124-149.  'minus' is the reverse complement, 'both' is seq followed
    by its reverse complement with NO separator (:141), so W-1 junction words exist.
    Non-ACGT symbols stay non-ACGT under complement (Biopython maps ambiguity codes to
    ambiguity codes); 'U' is treated as non-ACGT on both strands (parity unpinned there)."""
    if strand == "plus":
        return codes
    rc = codes[::-1].copy()
    ok = rc != INVALID
    rc[ok] ^= 1
    if strand == "minus":
        return rc
    if strand == "both":
        return np.concatenate([codes, rc])
    raise ValueError("strand must be one of both/plus/minus")


def pattern_info(pattern) -> tuple[str, int, int, list[int]]:
    """-k N gives the contiguous pattern '1'*N (phyloligo.py:1040-1041)."""
    if isinstance(pattern, (int, np.integer)):
        pattern = "1" * int(pattern)
    pattern = str(pattern)
    if not pattern or any(c not in "01" for c in pattern):
        raise ValueError("pattern must be a non-empty string of 0/1")
    ones = [i for i, c in enumerate(pattern) if c == "1"]
    return pattern, len(pattern), len(ones), ones


def count_pattern(codes: np.ndarray, pattern) -> tuple[np.ndarray, int]:
    """phyloligo.py:601-631 + the dense ordering of :653.  The reference splits the
    (upper-cased) sequence on runs of non-ACGT and slides a window of len(pattern) inside
    every piece that is long enough: a window counts iff ALL its positions (the '0'
    wildcard positions included) are A/C/G/T.  Returns (int64 counts[4^k], total)."""
    pattern, W, k, ones = pattern_info(pattern)
    D = 4 ** k
    L = codes.shape[0]
    if L < W:
        return np.zeros(D, dtype=np.int64), 0
    valid = (codes != INVALID).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(valid)])
    nwin = L - W + 1
    ok = (cs[W:W + nwin] - cs[:nwin]) == W
    idx = np.zeros(nwin, dtype=np.int64)
    for rank, pos in enumerate(ones):
        idx += codes[pos:pos + nwin].astype(np.int64) * (4 ** (k - 1 - rank))
    counts = np.bincount(idx[ok], minlength=D).astype(np.int64)
    return counts, int(ok.sum())


def count2freq(counts: np.ndarray, total: int) -> np.ndarray:
    """phyloligo.py:633-661: count/total in float64 (Python true division of two ints is
    correctly rounded, as is float64(count)/float64(total) for ints < 2^53); an empty
    record gives a row of zeros (:660)."""
    if total > 0:
        return counts.astype(np.float64) / np.float64(total)
    return np.zeros(counts.shape[0], dtype=np.float64)


def profile_counts(seq, pattern="1111", strand="both") -> tuple[np.ndarray, int]:
    """Integer half of compute_frequency (phyloligo.py:663-691)."""
    return count_pattern(select_strand(encode(seq), strand), pattern)


def profile_counts_per_window(seq, pattern="1111", strand="both") -> tuple[np.ndarray, int]:
    """The same integers the way the reference obtains them - one Python string per window (phyloligo.py:124-149 strand
    selection on the upper-cased text, :622-631 split at everything that is not A/C/G/T and join the pattern's '1' positions of
    every window, Counter; :653 the dense C,G,A,T word order).  Pure-Python loops: for small cases, and for the
    cpu_baseline leg of bench.py, which times THIS on a sample because count_pattern() above (numpy, one bincount per
    record) is ~15 x faster than what the reference actually runs per core (SURVEY: 2.87 s per 1 000 contigs of 2 kb)."""
    import collections
    import itertools
    import re
    pattern, W, k, ones = pattern_info(pattern)
    text = _as_bytes(seq).decode("latin-1").upper()
    if strand != "plus":
        rc = text[::-1].translate(str.maketrans("ACGT", "TGCA"))
        text = rc if strand == "minus" else text + rc
        if strand not in ("minus", "both"):
            raise ValueError("strand must be one of both/plus/minus")
    words: collections.Counter = collections.Counter()
    for piece in re.split("[^ACGT]+", text):
        for start in range(len(piece) - W + 1):
            words["".join(piece[start + o] for o in ones)] += 1
    dense = np.array([words.get("".join(w), 0) for w in itertools.product("CGAT", repeat=k)], dtype=np.int64)
    return dense, int(sum(words.values()))


def compute_frequency(seq, pattern="1111", strand="both") -> np.ndarray:
    """phyloligo.py:663-691."""
    counts, total = profile_counts(seq, pattern, strand)
    return count2freq(counts, total)


# ----------------------------------------------------------------------------------------
# FASTA ingest (phyloligo.py:869 -> Bio.SeqIO.parse(genome, "fasta"), third party, absent).
# Restated from Biopython's SimpleFastaParser: a record starts at a line beginning with
# '>', the sequence is the following lines each rstrip()-ed and joined, then ' ' and '\r'
# removed.  Blank lines before the first '>' are skipped, anything else there is an error.
# ----------------------------------------------------------------------------------------
def parse_fasta(data) -> tuple[list[str], list[bytes]]:
    if isinstance(data, str):
        with open(data, "rb") as fh:
            data = fh.read()
    titles: list[str] = []
    seqs: list[bytes] = []
    cur: list[bytes] | None = None
    for line in io.BytesIO(data):
        if line.startswith(b">"):
            if cur is not None:
                seqs.append(b"".join(cur).replace(b" ", b"").replace(b"\r", b""))
            titles.append(line[1:].rstrip().decode("latin-1"))
            cur = []
        elif cur is None:
            if line.strip():
                raise ValueError("FASTA data does not start with '>'")
        else:
            cur.append(line.rstrip())
    if cur is not None:
        seqs.append(b"".join(cur).replace(b" ", b"").replace(b"\r", b""))
    return titles, seqs


def compute_counts(seqs, pattern="1111", strand="both") -> tuple[np.ndarray, np.ndarray]:
    """Integer profile matrix of a list of sequences: (int64 counts[N, 4^k], int64 totals[N])."""
    _, _, k, _ = pattern_info(pattern)
    n = len(seqs)
    counts = np.zeros((n, 4 ** k), dtype=np.int64)
    totals = np.zeros(n, dtype=np.int64)
    for i, s in enumerate(seqs):
        counts[i], totals[i] = profile_counts(s, pattern, strand)
    return counts, totals


def counts_to_frequencies(counts: np.ndarray, totals: np.ndarray) -> np.ndarray:
    out = np.zeros(counts.shape, dtype=np.float64)
    nz = totals > 0
    out[nz] = counts[nz].astype(np.float64) / totals[nz].astype(np.float64)[:, None]
    return out


def compute_frequencies(seqs, pattern="1111", strand="both") -> np.ndarray:
    """phyloligo.py:847-877: one compute_frequency per record, vstack -> float64[N, 4^k]."""
    counts, totals = compute_counts(seqs, pattern, strand)
    return counts_to_frequencies(counts, totals)


# ----------------------------------------------------------------------------------------
# Per-pair metrics, phylodist.py.  Same numpy operations in the same order as the
# reference so that results are bit-identical where the reference is runnable.
# ----------------------------------------------------------------------------------------
def _posdef(d: np.ndarray) -> None:
    """phylodist.py:12-14: NaN and +-Inf entries become 0, in place."""
    d[np.isnan(d)] = 0
    d[np.isinf(d)] = 0


def KL(a: np.ndarray, b: np.ndarray) -> float:
    """phylodist.py:18-24 (1-D branch)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        d = a * np.log(a / b)
    _posdef(d)
    return np.sum(d)


def Eucl(a: np.ndarray, b: np.ndarray) -> float:
    """phylodist.py:36-41."""
    d = pow(a - b, 2)
    _posdef(d)
    return np.sqrt(np.sum(d))


def JSD(a: np.ndarray, b: np.ndarray) -> float:
    """phylodist.py:43-48 (1-D branch): natural log, no square root."""
    h = 0.5 * (a + b)
    return 0.5 * (KL(a, h) + KL(b, h))


def kendall_counts(x: np.ndarray, y: np.ndarray) -> tuple[int, int, int, int]:
    """con, dis, exx (tied in x only), exy (tied in y only) over all element pairs i<j,
    as the C Clustering Library's `kendall` counts them (Bio/Cluster/cluster.c, third
    party, absent from the tree; restated from its published algorithm)."""
    sx = np.sign(x[:, None] - x[None, :]).astype(np.int8)
    sy = np.sign(y[:, None] - y[None, :]).astype(np.int8)
    iu = np.triu_indices(x.shape[0], 1)
    sx, sy = sx[iu], sy[iu]
    prod = sx.astype(np.int64) * sy
    con = int((prod > 0).sum())
    dis = int((prod < 0).sum())
    exx = int(((sx == 0) & (sy != 0)).sum())
    exy = int(((sx != 0) & (sy == 0)).sum())
    return con, dis, exx, exy


def KT(a: np.ndarray, b: np.ndarray) -> float:
    """phylodist.py:71-74: 1 - Bio.Cluster.distancematrix((a,b), dist='k')[1][0].
    The library's Kendall distance is 1 - tau_b and is 1 when a denominator factor is 0,
    so KT is tau_b itself (a similarity: diagonal 1, constant row -> 0).
    SciPy-pinned (scipy.stats.kendalltau variant 'b'), not reference-pinned."""
    con, dis, exx, exy = kendall_counts(a, b)
    denomx = con + dis + exx
    denomy = con + dis + exy
    if denomx == 0 or denomy == 0:
        return 1.0 - 1.0
    tau = (con - dis) / np.sqrt(np.float64(denomx) * np.float64(denomy))
    return 1.0 - (1.0 - tau)


def BC(a: np.ndarray, b: np.ndarray) -> float:
    """'braycurtis' string metric of phyloligo.py:381 -> SciPy's C loop: sum|a-b| / sum|a+b|,
    both sums accumulated sequentially (cumsum reproduces that order bit for bit)."""
    if a.shape[0] == 0:
        return np.float64(np.nan)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.cumsum(np.abs(a - b))[-1] / np.cumsum(np.abs(a + b))[-1]


def rank_average(x: np.ndarray) -> np.ndarray:
    """Average ranks (1-based), ties share the mean rank (scipy.stats.rankdata default)."""
    order = np.argsort(x, kind="mergesort")
    xs = x[order]
    n = x.shape[0]
    ranks = np.empty(n, dtype=np.float64)
    i = 0
    while i < n:
        j = i
        while j + 1 < n and xs[j + 1] == xs[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return ranks


def SC(a: np.ndarray, b: np.ndarray) -> float:
    """phylodist.py:82-85 as intended: 1 - spearmanr(a, b).correlation (the reference
    raises NameError as shipped).  SciPy-pinned.  A constant row gives NaN."""
    ra = rank_average(a) - (a.shape[0] + 1) / 2.0
    rb = rank_average(b) - (b.shape[0] + 1) / 2.0
    with np.errstate(divide="ignore", invalid="ignore"):
        rho = np.float64(np.dot(ra, rb)) / np.sqrt(np.float64(np.dot(ra, ra)) * np.float64(np.dot(rb, rb)))
    return 1.0 - rho


_PAIR = {"Eucl": Eucl, "JSD": JSD, "KT": KT, "BC": BC, "SC": SC}


def pairwise_distances(freq: np.ndarray, metric: str = "Eucl") -> np.ndarray:
    """compute_distances_joblib (phyloligo.py:364-392) with n_jobs=1, i.e. sklearn's
    pairwise_distances: callables -> upper triangle, mirrored, then the diagonal from
    metric(x, x); 'braycurtis' -> squareform(pdist(X)) whose diagonal is exactly 0."""
    if metric not in _PAIR:
        raise ValueError("unknown metric %r" % (metric,))
    f = np.asarray(freq, dtype=np.float64)
    n = f.shape[0]
    fn = _PAIR[metric]
    out = np.zeros((n, n), dtype=np.float64)
    for i in range(n):
        for j in range(i + 1, n):
            out[i, j] = fn(f[i], f[j])
    out = out + out.T
    if metric != "BC":
        for i in range(n):
            out[i, i] = fn(f[i], f[i])
    return out


def pairwise_rows(freq: np.ndarray, metric: str, rows) -> np.ndarray:
    """Rows `rows` x all columns of pairwise_distances without the O(N^2) loop: used for
    spot checks at large N and for the bounded cpu_baseline sample of bench.py."""
    f = np.asarray(freq, dtype=np.float64)
    fn = _PAIR[metric]
    rows = list(rows)
    out = np.zeros((len(rows), f.shape[0]), dtype=np.float64)
    for r, i in enumerate(rows):
        for j in range(f.shape[0]):
            if i == j and metric == "BC":
                out[r, j] = 0.0
            else:
                out[r, j] = fn(f[i], f[j])
    return out


# ----------------------------------------------------------------------------------------
# Block (vectorised) forms: same mathematics, numpy reductions over an axis.  Used where
# the per-pair Python loop would take minutes; agree with the per-pair forms to ~1e-15.
# ----------------------------------------------------------------------------------------
def pairwise_block(freq: np.ndarray, metric: str, row_begin: int = 0, row_end: int | None = None) -> np.ndarray:
    f = np.asarray(freq, dtype=np.float64)
    n = f.shape[0]
    row_end = n if row_end is None else row_end
    a = f[row_begin:row_end]
    out = np.empty((a.shape[0], n), dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        for r in range(a.shape[0]):
            x = a[r][None, :]
            if metric == "Eucl":
                out[r] = np.sqrt(((x - f) ** 2).sum(axis=1))
            elif metric == "JSD":
                h = 0.5 * (x + f)
                t1 = x * np.log(x / h)
                t1[~np.isfinite(t1)] = 0
                t2 = f * np.log(f / h)
                t2[~np.isfinite(t2)] = 0
                out[r] = 0.5 * (t1.sum(axis=1) + t2.sum(axis=1))
            elif metric == "BC":
                out[r] = np.abs(x - f).sum(axis=1) / np.abs(x + f).sum(axis=1)
                if row_begin + r < n:
                    out[r, row_begin + r] = 0.0
            elif metric in ("KT", "SC"):
                fn = _PAIR[metric]
                for j in range(n):
                    out[r, j] = fn(a[r], f[j])
            else:
                raise ValueError("unknown metric %r" % (metric,))
    return out


# ----------------------------------------------------------------------------------------
# Output text, phyloligo.py:1059-1066: numpy.savetxt(path, M, delimiter="\t") -> "%.18e".
# ----------------------------------------------------------------------------------------
def mat_bytes(m: np.ndarray) -> bytes:
    buf = io.BytesIO()
    np.savetxt(buf, np.asarray(m), delimiter="\t")
    return buf.getvalue()


# ----------------------------------------------------------------------------------------
# Synthetic assemblies of SURVEY.md section 8d / BASELINE.md section 4.
# ----------------------------------------------------------------------------------------
def synthetic_contigs(n: int, length: int = 2000, seed: int = 1001) -> list[bytes]:
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [lut[rng.integers(0, 4, size=length, dtype=np.uint8)].tobytes() for _ in range(n)]


def fasta_bytes(seqs, width: int = 80) -> bytes:
    out = []
    for i, s in enumerate(seqs):
        out.append(b">c%07d\n" % i)
        for p in range(0, len(s), width):
            out.append(s[p:p + width] + b"\n")
    return b"".join(out)

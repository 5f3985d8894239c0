#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN FUNCTIONS.

Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference cannot be imported as shipped here: Biopython, scoop and h5py are absent and
`sklearn.externals.joblib` no longer exists (ordinary ModuleNotFoundError).
This harness registers minimal stand-in modules for those imports and then loads
/root/reference/phylopackage/bin/phyloligo.py with importlib, so that the functions below
are the reference's code, executed unmodified:

    ref.compute_frequency, ref.cut_sequence_and_count_pattern, ref.count2freq,
    ref.compute_distances_joblib (Eucl, JSD, BC), phylodist.Eucl/JSD/KL, numpy.savetxt.

Only numbers (inputs and the reference's outputs) are stored.  The stand-in `Seq` class
complements A/C/G/T/N in both cases and leaves every other symbol unchanged.  What that
leaves stand-in dependent on the 'minus' / 'both' strands is exactly ONE character class:
  * IUPAC ambiguity codes (RYKMSWBDHVN, any case): Biopython maps them to ambiguity codes,
    the stand-in leaves them as they are -- not A/C/G/T either way, so the split on
    [^ACGT]+ (bin/phyloligo.py:625) gives the same words: stand-in INDEPENDENT;
  * gaps, digits, '*', '.', any other byte: unchanged by both: stand-in INDEPENDENT;
  * 'U' / 'u': the stand-in leaves it (a separator on both strands); Biopython's DNA
    complement table maps U -> A (releases that do not raise on mixed T/U), which would
    make the minus-strand copy of a U countable as 'A': stand-in DEPENDENT, unpinned.
    (The plus strand is pinned: 'U' is not in ACGT, bin/phyloligo.py:625.)
KT and SC vectors come from SciPy and are labelled `scipy_`.
"""
import importlib.util
import io
import os
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def load_reference():
    import joblib

    _stub("scoop", futures=types.SimpleNamespace(map=map))
    _stub("h5py")
    table = str.maketrans("ACGTacgtNn", "TGCAtgcaNn")

    class Seq(str):
        def reverse_complement(self):
            return Seq(str(self).translate(table)[::-1])

    bio = _stub("Bio")
    bio.Seq = _stub("Bio.Seq", Seq=Seq)
    bio.SeqIO = _stub("Bio.SeqIO")
    bio.Cluster = _stub("Bio.Cluster")
    import sklearn.externals as ext   # the real package; only its long-gone `joblib` submodule is stood in
    ext.joblib = _stub("sklearn.externals.joblib", Parallel=joblib.Parallel, delayed=joblib.delayed,
                       dump=joblib.dump, load=joblib.load)
    sys.path.insert(0, REF_ROOT)
    spec = importlib.util.spec_from_file_location(
        "phyloligo_ref", os.path.join(REF_ROOT, "phylopackage/bin/phyloligo.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


PATTERNS = ["1", "11", "1111", "11111", "1101", "10011", "110101", "11011011"]
STRANDS = ["both", "plus", "minus"]
# W = 33 (k = 2), W = 41 (k = 4, not a palindrome), W = 64 (k = 6, the widest window the library takes)
WIDE_PATTERNS = ["1" + "0" * 31 + "1", "1101" + "0" * 36 + "1", "11" + "0" * 30 + "101" + "0" * 27 + "11"]


def build_contigs():
    """Hand-built edge cases + seeded random contigs (lengths 0..5000, N runs, lower case, IUPAC)."""
    rng = np.random.default_rng(20261003)
    hand = [
        "", "A", "AC", "ACG", "ACGT", "ACGTC", "ACNGT", "NNNNNNNN", "acgtn", "ACGTACGTACGTACGTACGTACGT",
        "CCCCCCCCCCCC", "TTTTTTTTTTTTTTTT", "ACGTNACGTNNACGTTGCA", "acgtACGTacgtRYKMacgtSWBDHVNacgt",
        "ACGT-ACGT*ACGT.ACGT1234ACGT", "GATTACAGATTACAGATTACA", "NACGTACGTACGTN", "ACGTACGNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNACGTTTGA",
        "TGCATGCATGCATGCATGCATGCAUUUUTGCATGCA".replace("U", "X"), "CGCGCGCGCGCGCGCGCGCGCGCGCGCGCGCG",
        "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAC", "ACGTAC", "ACGTACG", "ACGTACGT", "ACGTACGTA", "nnnnACGTACGTnnnn",
    ]
    out = list(hand)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    for length in [5, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 500,
                   777, 1000, 1023, 1024, 1025, 2000, 2048, 3000, 4096, 5000]:
        s = alpha[rng.integers(0, 4, size=length)].copy()
        out.append(s.tobytes().decode())
    for length in [50, 200, 800, 2000, 3500]:           # N runs + lower case + IUPAC sprinkled in
        s = alpha[rng.integers(0, 4, size=length)].copy()
        for _ in range(max(1, length // 100)):
            p = int(rng.integers(0, length))
            q = min(length, p + int(rng.integers(1, 12)))
            s[p:q] = ord("N")
        for _ in range(max(1, length // 50)):
            s[int(rng.integers(0, length))] = int(rng.choice(np.frombuffer(b"RYKMSWBDHVn", dtype=np.uint8)))
        low = rng.random(length) < 0.3
        s[low] = np.char.lower(s[low].view("S1")).view(np.uint8)
        out.append(s.tobytes().decode())
    skew = np.frombuffer(b"AAAACGTT", dtype=np.uint8)      # compositionally different contigs
    for length in [600, 1500, 2500]:
        out.append(skew[rng.integers(0, 8, size=length)].tobytes().decode())
    return out


def main():
    ref = load_reference()
    from phylopackage import phylodist  # the reference's metric library, via sys.path above

    contigs = build_contigs()
    store = {"contigs": np.array([c.encode() for c in contigs], dtype="S")}

    # (1) profiles: the reference's Counter -> dense vector, for every pattern x strand
    for pat in PATTERNS:
        k = pat.count("1")
        words = ["".join(w) for w in ref.product(("C", "G", "A", "T"), repeat=k)]
        for strand in STRANDS:
            counts = np.zeros((len(contigs), 4 ** k), dtype=np.int64)
            totals = np.zeros(len(contigs), dtype=np.int64)
            freqs = np.zeros((len(contigs), 4 ** k), dtype=np.float64)
            for i, s in enumerate(contigs):
                strand_seq = ref.select_strand(s, strand).upper()
                cw, tot = ref.cut_sequence_and_count_pattern(strand_seq, pat)
                counts[i] = [cw.get(w, 0) for w in words]
                totals[i] = tot
                freqs[i] = ref.compute_frequency(s, pat, strand)
            store["counts_%s_%s" % (pat, strand)] = counts
            store["totals_%s_%s" % (pat, strand)] = totals
            store["freq_%s_%s" % (pat, strand)] = freqs
    np.savez_compressed(os.path.join(HERE, "profiles.npz"), **store)

    # (1b) spaced seeds wider than 32 positions (33..64): same contigs, same reference functions
    wide = {"contigs": store["contigs"]}
    for pat in WIDE_PATTERNS:
        k = pat.count("1")
        words = ["".join(w) for w in ref.product(("C", "G", "A", "T"), repeat=k)]
        for strand in STRANDS:
            counts = np.zeros((len(contigs), 4 ** k), dtype=np.int64)
            totals = np.zeros(len(contigs), dtype=np.int64)
            for i, s in enumerate(contigs):
                cw, tot = ref.cut_sequence_and_count_pattern(ref.select_strand(s, strand).upper(), pat)
                counts[i] = [cw.get(w, 0) for w in words]
                totals[i] = tot
            wide["counts_%s_%s" % (pat, strand)] = counts
            wide["totals_%s_%s" % (pat, strand)] = totals
    np.savez_compressed(os.path.join(HERE, "profiles_wide.npz"), **wide)

    # (2) distance matrices of a 48-contig set through the reference's joblib driver
    rng = np.random.default_rng(48)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    set48 = [alpha[rng.integers(0, 4, size=int(rng.integers(300, 2500)))].tobytes().decode() for _ in range(44)]
    set48 += [set48[3], "", "NNNNNNNN", set48[10][:900]]      # a duplicate, two empty profiles, a prefix
    dist = {"contigs": np.array([c.encode() for c in set48], dtype="S")}
    for pat, strand in [("1111", "both"), ("11", "plus"), ("1101", "minus"), ("11011011", "both")]:
        freq = np.vstack([ref.compute_frequency(s, pat, strand) for s in set48])
        key = "%s_%s" % (pat, strand)
        dist["freq_" + key] = freq
        for metric in ("Eucl", "JSD", "BC"):
            m = ref.compute_distances_joblib(freq, metric=metric, n_jobs=1)
            dist["%s_%s" % (metric, key)] = m
        if pat == "1111":
            buf = io.BytesIO()
            np.savetxt(buf, dist["JSD_" + key], delimiter="\t")
            dist["matbytes_JSD_" + key] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
            # per-pair functions called directly (phylodist.py), first 8 rows
            sub = freq[:8]
            dist["pair_Eucl_" + key] = np.array([[phylodist.Eucl(a, b) for b in sub] for a in sub])
            dist["pair_JSD_" + key] = np.array([[phylodist.JSD(a, b) for b in sub] for a in sub])
            dist["pair_KL_" + key] = np.array([[phylodist.KL(a, b) for b in sub] for a in sub])
        # SciPy-pinned vectors for the two metrics the reference cannot run here
        from scipy.stats import kendalltau, spearmanr
        n = 16 if pat != "11011011" else 6
        kt = np.zeros((n, n))
        sc = np.zeros((n, n))
        with np.errstate(all="ignore"):
            for i in range(n):
                for j in range(n):
                    kt[i, j] = kendalltau(freq[i], freq[j], variant="b").statistic
                    sc[i, j] = 1.0 - spearmanr(freq[i], freq[j]).correlation
        dist["scipy_KT_" + key] = kt
        dist["scipy_SC_" + key] = sc
    np.savez_compressed(os.path.join(HERE, "distances.npz"), **dist)

    # (3) C1 synthetic set (seed 1001, 1000 x 2 kb): hash of the counts + 16x16 corners
    rng = np.random.default_rng(1001)
    c1 = [alpha[rng.integers(0, 4, size=2000, dtype=np.uint8)].tobytes().decode() for _ in range(1000)]
    freq = np.vstack([ref.compute_frequency(s, "1111", "both") for s in c1])
    c1store = {"freq_first16": freq[:16], "freq_colsum": freq.sum(axis=0), "freq_rowsum": freq.sum(axis=1)}
    for metric in ("Eucl", "JSD", "BC"):
        c1store["corner_" + metric] = ref.compute_distances_joblib(freq[:16], metric=metric, n_jobs=1)
        c1store["far_" + metric] = np.array([[ref.compute_distances_joblib(
            np.vstack([freq[i], freq[j]]), metric=metric, n_jobs=1)[0, 1] for j in (500, 777, 999)] for i in (0, 1, 2)])
    np.savez_compressed(os.path.join(HERE, "c1_synthetic.npz"), **c1store)

    # (4) CLI resolution of -k / -p (phyloligo.py:1000-1041)
    cli = {}
    for name, argv in {
        "default": ["-i", "x.fa", "--method", "joblib"],
        "k6": ["-i", "x.fa", "--method", "joblib", "-k", "6"],
        "p": ["-i", "x.fa", "--method", "joblib", "-p", "11011"],
        "k_then_p": ["-i", "x.fa", "--method", "joblib", "-k", "5", "-p", "101"],
        "p_then_k": ["-i", "x.fa", "--method", "joblib", "-p", "101", "-k", "5"],
    }.items():
        old = sys.argv
        sys.argv = ["phyloligo.py"] + argv
        try:
            params = ref.get_cmd()
        finally:
            sys.argv = old
        pat = params.pattern
        if type(pat) == int:
            pat = "1" * pat
        cli[name] = np.array([" ".join(argv), str(pat), params.strand, params.dist, params.out_file, params.large,
                              str(params.threads_max)], dtype="U")
    np.savez_compressed(os.path.join(HERE, "cli.npz"), **cli)
    # (5) FASTA cases: everything DOWNSTREAM of SeqIO.parse is the reference's own code.  Bio.SeqIO is absent here, so
    # the records come from the oracle's restated parser (oracle.parse_fasta: that restatement stays unpinned); every
    # parsed string then goes through ref.select_strand / cut_sequence_and_count_pattern / compute_frequency unmodified.
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import phyloligo_oracle as oracle
    from tests.fasta_cases import CASES, PROFILE_KEYS
    fa = {}
    for name, data in sorted(CASES.items()):
        titles, seqs = oracle.parse_fasta(data)
        fa["data_" + name] = np.frombuffer(data, dtype=np.uint8)
        fa["titles_" + name] = np.array([t.encode("latin-1") for t in titles], dtype="S") if titles else np.zeros(0, dtype="S1")
        fa["seqs_" + name] = np.array(seqs, dtype="S") if any(seqs) else np.array([b""] * len(seqs), dtype="S1")
        fa["seqlens_" + name] = np.array([len(x) for x in seqs], dtype=np.int64)
        for pat, strand in PROFILE_KEYS:
            k = pat.count("1")
            words = ["".join(w) for w in ref.product(("C", "G", "A", "T"), repeat=k)]
            counts = np.zeros((len(seqs), 4 ** k), dtype=np.int64)
            totals = np.zeros(len(seqs), dtype=np.int64)
            freqs = np.zeros((len(seqs), 4 ** k), dtype=np.float64)
            for i, sq in enumerate(seqs):
                text = sq.decode("latin-1")
                cw, tot = ref.cut_sequence_and_count_pattern(ref.select_strand(text, strand).upper(), pat)
                counts[i] = [cw.get(w, 0) for w in words]
                totals[i] = tot
                freqs[i] = ref.compute_frequency(text, pat, strand)
            fa["counts_%s_%s_%s" % (name, pat, strand)] = counts
            fa["totals_%s_%s_%s" % (name, pat, strand)] = totals
            fa["freq_%s_%s_%s" % (name, pat, strand)] = freqs
    np.savez_compressed(os.path.join(HERE, "fasta_cases.npz"), **fa)
    # (6) SURVEY 8f-1: the --large memmap variant through the reference's OWN memmap code (SURVEY 8c
    # listed it as "not pinnable" because the module needs h5py / scoop to import - the stand-ins above solve that, and
    # compute_distances_memmap itself needs only numpy + joblib + sklearn).  Frequencies go into a float32 memmap exactly as
    # compute_frequencies_joblib_memmap fills it (bin/phyloligo.py:903-913: compute_frequency_memmap per record), then
    # compute_distances_memmap (:394-427) -> euclidean_distances_loc / JSD_loc (:200-207) writes the float32 container.
    # The bytes of that container are the fixture.  BC_loc / KT_loc / SC_loc are broken in the reference (:209-222): what
    # they raise is recorded, not worked around.  (n_jobs = 1: loky workers cannot unpickle functions of a module loaded
    # from a file path under a stand-in name; the workers' row slices are disjoint, so the bytes do not depend on it.)
    import tempfile
    mm = {"contigs": dist["contigs"]}
    with tempfile.TemporaryDirectory() as tmp:
        for pat, strand in [("1111", "both"), ("1101", "minus")]:
            key = "%s_%s" % (pat, strand)
            for metric in ("Eucl", "JSD", "BC", "KT", "SC"):
                folder = tempfile.mkdtemp(dir=tmp)                 # compute_distances_memmap removes dirname(freq_name) (:426-427)
                freq_name = os.path.join(folder, "frequencies")
                fr = np.memmap(freq_name, dtype=np.float32, shape=(len(set48), 4 ** pat.count("1")), mode="w+")
                for i, sq in enumerate(set48):
                    ref.compute_frequency_memmap(fr, i, sq, pat, strand)
                mm["freq32_" + key] = np.array(fr)
                out = os.path.join(tmp, "dist_%s_%s.f32" % (metric, key))
                try:
                    ref.compute_distances_memmap(fr, freq_name, out, metric=metric, n_jobs=1)
                except Exception as exc:                          # noqa: BLE001 - the reference's own failure is the datum
                    mm["raises_%s_%s" % (metric, key)] = np.array(["%s: %s" % (type(exc).__name__, exc)], dtype="U")
                    continue
                finally:
                    del fr
                mm["container_%s_%s" % (metric, key)] = np.fromfile(out, dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "memmap.npz"), **mm)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()

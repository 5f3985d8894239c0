#!/usr/bin/env python3
"""Generate tests/golden/kount.npz by RUNNING THE REFERENCE'S OWN Kount.py functions.

Build container only (needs /root/reference):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_kount.py

Kount.py imports Biopython (absent here): `Bio.SeqIO.parse` and `Bio.Seq.Seq` are stood in by a
minimal FASTA reader / reverse-complement class (same stand-ins as make_golden.py, plus a record
type with .seq and .id), `scoop` by an empty module, `sklearn.externals.joblib` by the real joblib.  Everything computed below --
window cutting and displayed coordinates (make_genome_chunk), the N gate and profiles
(compute_frequency), the whole-genome prototype (compute_whole_composition), the distances
(compute_distances -> JSD / KL / Eucl) and the `.dist` file main() writes -- is the reference's code.
"""
import importlib.util
import io
import os
import sys
import tempfile
import types

import numpy as np

REF_ROOT = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def load_kount():
    import joblib
    table = str.maketrans("ACGTacgtNn", "TGCAtgcaNn")

    class Seq(str):
        def reverse_complement(self):
            return Seq(str(self).translate(table)[::-1])

    class Record:
        def __init__(self, title, seq):
            self.id = title.split(None, 1)[0] if title.split() else ""
            self.seq = Seq(seq)

    def parse(path, fmt):
        title, chunks = None, []
        with open(path) as fh:
            for line in fh:
                if line.startswith(">"):
                    if title is not None:
                        yield Record(title, "".join(chunks).replace(" ", "").replace("\r", ""))
                    title, chunks = line[1:].rstrip(), []
                elif title is not None:
                    chunks.append(line.rstrip())
        if title is not None:
            yield Record(title, "".join(chunks).replace(" ", "").replace("\r", ""))

    _stub("scoop", futures=types.SimpleNamespace(map=map))
    bio = _stub("Bio")
    bio.Seq = _stub("Bio.Seq", Seq=Seq)
    bio.SeqIO = _stub("Bio.SeqIO", parse=parse)
    import sklearn.externals as ext
    ext.joblib = _stub("sklearn.externals.joblib", Parallel=joblib.Parallel, delayed=joblib.delayed,
                       dump=joblib.dump, load=joblib.load)
    spec = importlib.util.spec_from_file_location("kount_ref", os.path.join(REF_ROOT, "phylopackage/bin/Kount.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def build_genome(rng):
    """Records in each of make_genome_chunk's three regimes for -w 1000 -t 200, with N runs and lower case."""
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for name, length in (("short1 first", 600), ("mid1", 2500), ("long1 desc here", 9000), ("short2", 999),
                         ("edge1000", 1000), ("edge1200", 1200), ("mid2", 3999), ("long2", 4000), ("long3", 12345)):
        s = alpha[rng.integers(0, 4, size=length)].copy()
        recs.append([name, s])
    # N runs: one window mostly N (gated at -n 0.4), one partly N (kept), lower-case stretch
    recs[2][1][3000:3700] = ord("N")
    recs[2][1][5200:5350] = ord("N")
    recs[8][1][100:900] = ord("N")
    low = recs[8][1][6000:7000]
    recs[8][1][6000:7000] = np.char.lower(low.view("S1")).view(np.uint8)
    recs[1][1][400:460] = ord("n")          # lower-case n is NOT counted by the gate (seq.count('N')), but breaks words
    out = io.StringIO()
    for name, s in recs:
        out.write(">%s\n" % name)
        txt = s.tobytes().decode()
        for p in range(0, len(txt), 70):
            out.write(txt[p:p + 70] + "\n")
    return out.getvalue()


def main():
    ref = load_kount()
    rng = np.random.default_rng(424242)
    genome_txt = build_genome(rng)
    conta_txt = ">c1\n" + "".join("ACGGT"[i % 5] for i in range(3000)) + "\n>c2\n" + "".join("AATTC"[(i * i) % 5] for i in range(2000)) + "\n"
    store = {"genome_fasta": np.frombuffer(genome_txt.encode(), dtype=np.uint8),
             "conta_fasta": np.frombuffer(conta_txt.encode(), dtype=np.uint8)}
    with tempfile.TemporaryDirectory() as tmp:
        gpath, cpath = os.path.join(tmp, "genome.fa"), os.path.join(tmp, "conta.fa")
        open(gpath, "w").write(genome_txt)
        open(cpath, "w").write(conta_txt)
        opts = types.SimpleNamespace(strand="both", threads_max=1, n_max_freq_in_windows=0.4)
        for pattern, strand in (("1111", "both"), ("11", "plus"), ("1111", "minus")):
            opts.strand = strand
            key = "%s_%s" % (pattern, strand)
            proto = ref.compute_whole_composition(gpath, pattern, strand, nb_jobs=1)
            store["proto_" + key] = np.asarray(proto, dtype=np.float64)
            infos, seqs = [], []
            for chunk_info, sequences in ref.make_genome_chunk(gpath, 1000, 200, opts, 50000):
                infos += chunk_info
                seqs += sequences
            store["win_id_" + key] = np.array([i[0] for i in infos], dtype="U")
            store["win_start_" + key] = np.array([i[1] for i in infos], dtype=np.int64)
            store["win_stop_" + key] = np.array([i[2] for i in infos], dtype=np.int64)
            store["win_len_" + key] = np.array([len(s) for s in seqs], dtype=np.int64)
            for dist in ("JSD", "KL", "Eucl"):
                vec = ref.compute_distances("joblib", "None", dist, proto, seqs, pattern, strand, 1, 0.4)
                store["dist_%s_%s" % (dist, key)] = np.asarray(vec, dtype=np.float64)
        # full CLI runs: genome windows vs whole genome, and with a contaminant training set
        for name, argv in {"whole_JSD": ["-i", gpath, "-w", "1000", "-t", "200", "-d", "JSD", "-W", os.path.join(tmp, "o1")],
                           "whole_Eucl_k2": ["-i", gpath, "-w", "1000", "-t", "200", "-d", "Eucl", "-k", "2", "-s", "plus", "-W", os.path.join(tmp, "o2")],
                           "conta_KL": ["-i", gpath, "-c", cpath, "-w", "1500", "-t", "300", "-d", "KL", "-W", os.path.join(tmp, "o3")]}.items():
            old = sys.argv
            sys.argv = ["Kount.py"] + argv
            try:
                ref.main()
            except SystemExit:
                pass
            finally:
                sys.argv = old
            outdir = argv[-1]
            for fn in sorted(os.listdir(outdir)):
                store["cli_%s__%s" % (name, fn)] = np.frombuffer(open(os.path.join(outdir, fn), "rb").read(), dtype=np.uint8)
            store["cli_%s__argv" % name] = np.array([a.replace(tmp, "TMP") for a in argv], dtype="U")
    np.savez_compressed(os.path.join(HERE, "kount.npz"), **store)
    print("kount golden written:", sorted(k for k in store if k.startswith("cli_")))


if __name__ == "__main__":
    main()

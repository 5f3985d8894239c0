#!/usr/bin/env python3
"""Host-side mirror of Kount.py, the sliding-window front end of ContaLocate, on the HIP path.

Same options, window rules, N gate, x1000 display scaling and `.dist` lines as
/root/reference/phylopackage/bin/Kount.py (get_cmd :483-520, main :522-603, make_genome_chunk :343-407,
compute_frequency :274-301, JSD/KL/Eucl :69-123).  The windows are byte ranges of the one sequence buffer in
HBM (no per-window strings): po_count_profiles_ranges counts them all in one launch, po_profile_distances
compares every window with the prototype profile.  No CPU fallback.
"""
import argparse
import os
import sys

import numpy as np

from . import api
from .phyloligo import _context, read_fasta, read_fasta_device

MIN_WINDOWS_PARALLEL = 20      # min_nb_w_per_fasta_for_mul_cpu, Kount.py:64: switches the coordinate rules

_genomes = {}                  # (path, mtime, size) -> parsed FASTA + its sequence bytes resident in HBM


def _load_genome(genome):
    """The reference parses the assembly once per prototype and once per scan (Kount.py:306, :417); here the parsed
    records and the device copy of the sequence bytes are kept per file, so every later step starts from HBM."""
    import torch
    st = os.stat(genome)
    key = (os.path.abspath(genome), st.st_mtime_ns, st.st_size)
    hit = _genomes.get(key)
    if hit is None:
        ingest = read_fasta_device(genome)
        if ingest is not None:                                # parsed in HBM; only the record offsets come back
            d_seq, d_off, titles = ingest
            n_bytes = d_seq.numel()
            offsets = d_off.cpu().numpy().astype(np.uint64)
            d_seq = d_seq._base if d_seq._base is not None else d_seq      # the 16-byte padded allocation
        else:
            seq, offsets, titles = read_fasta(genome)
            dev = torch.device("cuda", _context().device)
            n_bytes = len(seq)
            pad = (-len(seq)) % 16                            # the device buffer is 16-byte aligned and padded
            d_seq = torch.from_numpy(np.concatenate([seq, np.zeros(pad, np.uint8)]) if pad else seq).to(dev)
            d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
        _genomes.clear()                                      # one assembly at a time
        hit = _genomes[key] = (n_bytes, offsets, titles, d_seq, d_off)
    return hit


def record_windows(length, wsize, wstep):
    """(start, displayed_start, displayed_stop) of every window of one record (Kount.py:352-401): a record
    shorter than the window is one window; below 20 steps the 'serial' display rule applies, else the
    'parallel' one.  Windows start at range(0, len - wsize, wstep)."""
    s, d0, d1 = record_windows_arrays(length, wsize, wstep)
    return list(zip(s.tolist(), d0.tolist(), d1.tolist()))


def record_windows_arrays(length, wsize, wstep):
    """The same as three int64 arrays (a 20 Mb chromosome has 40 000 windows: no Python loop over them)."""
    length = int(length)
    if length < wsize:
        return np.zeros(1, np.int64), np.zeros(1, np.int64), np.full(1, length, np.int64)
    half_lo, half_hi = wsize / 2 - wstep / 2, wsize / 2 + wstep / 2
    s = np.arange(0, length - wsize, wstep, dtype=np.int64)
    if length < MIN_WINDOWS_PARALLEL * wstep:
        d0 = np.where(s == 0, 1, (s + half_lo).astype(np.int64))            # int(): truncation of a non-negative float
        d1 = np.where(s == length - wsize, length, (s + half_hi).astype(np.int64))
    else:
        start, stop = (s + half_lo).astype(np.int64), (s + half_hi).astype(np.int64)
        edge = stop - wstep / 2 + wsize / 2
        d0 = np.where(start == half_lo, 1, start)
        d1 = np.where((length - wstep <= edge) & (edge <= length), length, stop)
    return s, d0.astype(np.int64), d1.astype(np.int64)


def compute_whole_composition(genome, pattern, strand, nb_jobs=1):
    """Kount.py:303-319: word counts summed over every record, then one frequency vector."""
    n_bytes, offsets, _, d_seq, d_off = _load_genome(genome)
    counts, totals = _context().count_profiles(d_seq[:n_bytes], d_off, pattern, strand)
    c = counts.sum(dim=0, dtype=__import__("torch").int64).cpu().numpy()
    t = int(totals.sum().item())
    return c.astype(np.float64) / np.float64(t) if t > 0 else np.zeros(c.shape[0], dtype=np.float64)


def sliding_windows_distances(genome, mcp_comparison, mth_dist="JSD", pattern="1111", windows_size=5000,
                              windows_step=500, options=None):
    """Kount.py:409-458, all windows at once: list of [id, displayed_start, displayed_stop, distance]."""
    strand = getattr(options, "strand", "both")
    n_max = float(getattr(options, "n_max_freq_in_windows", 0.4))
    import torch
    n_bytes, offsets, titles, d_seq, _ = _load_genome(genome)
    ids, begins, ends, d0s, d1s = [], [], [], [], []
    for r, title in enumerate(titles):
        off, length = int(offsets[r]), int(offsets[r + 1] - offsets[r])
        parts = title.split(None, 1)
        rid = parts[0] if parts else ""
        s, d0, d1 = record_windows_arrays(length, windows_size, windows_step)
        ids.extend([rid] * len(s))
        begins.append(off + s)
        ends.append(off + np.minimum(length, s + windows_size))
        d0s.append(d0)
        d1s.append(d1)
    if not ids:
        return []
    begins, ends = np.concatenate(begins), np.concatenate(ends)
    d0s, d1s = np.concatenate(d0s).tolist(), np.concatenate(d1s).tolist()
    ctx = _context()
    # windows, their profiles and their distances stay in HBM; 8 + 8 bytes per window come back
    d_begins, d_ends = torch.from_numpy(begins).to(d_seq.device), torch.from_numpy(ends).to(d_seq.device)
    counts, totals = ctx.count_profiles_ranges(d_seq[:n_bytes], d_begins, d_ends, pattern, strand)
    raw = ctx.profile_distances(counts, totals, np.asarray(mcp_comparison, dtype=np.float64), mth_dist).cpu().numpy()
    dist = raw if mth_dist == "KL" else raw * 1000                        # display scaling of JSD / Eucl (:96, :123)
    # the N gate (:295-300): proportion of upper-case 'N' in the window
    n_in_window = ctx.count_byte_ranges(d_seq, d_begins, d_ends, ord("N")).cpu().numpy()
    lens = ends - begins
    nfrac = n_in_window / np.maximum(lens, 1)
    gated = (lens > 0) & (nfrac > n_max)
    if gated.any():
        k = api.normalise_pattern(pattern).count("1")
        if k ** 4 != 4 ** k:      # the reference builds a NaN vector of length k**4 (:300), which only fits k = 2, 4
            raise ValueError("operands could not be broadcast together with shapes (%d,) (%d,)" % (k ** 4, 4 ** k))
        dist = np.where(gated, 0.0, dist)                                 # NaN profile: every term is dropped
    return [list(t) for t in zip(ids, d0s, d1s, dist)]


def get_cmd(argv=None):
    """Kount.py:483-520, option for option."""
    parser = argparse.ArgumentParser(prog="Kount.py")
    parser.add_argument("-i", "--assembly", action="store", required=True, dest="genome",
                        help="multifasta of the genome assembly")
    parser.add_argument("-c", "--conta", action="store", dest="conta",
                        help="multifasta of the contaminant species training set")
    parser.add_argument("-r", "--host", action="store", dest="host",
                        help="optional host species training set in multifasta")
    parser.add_argument("-n", "--n_max_freq_in_windows", action="store", type=float, dest="n_max_freq_in_windows",
                        default=0.4, help="maximum proportion of N tolerated in a window [0~1]")
    parser.add_argument("-k", "--lgMot", action="store", dest="k", type=int, default=4,
                        help="word wise/ kmer lenght/ k [default:%(default)d]")
    parser.add_argument("-p", "--pattern", action="store", dest="pattern", help="pattern to use for frequency computation")
    parser.add_argument("-w", "--windows_size", action="store", dest="windows_size", type=int, default=5000,
                        help="Sliding windows size (bp)")
    parser.add_argument("-t", "--windows_step", action="store", dest="windows_step", type=int, default=500,
                        help="Sliding windows step size(bp)")
    parser.add_argument("-d", "--distance", action="store", dest="dist", choices=["JSD", "Eucl", "KL"], default="JSD",
                        help="distance method between two signatures [default:%(default)s]")
    parser.add_argument("-s", "--strand", action="store", default="both", choices=["both", "plus", "minus"],
                        help="strand used to compute microcomposition. [default:%(default)s]")
    parser.add_argument("-u", "--cpu", action="store", dest="threads_max", type=int, default=4,
                        help="accepted for compatibility (host threads are not the compute resource)")
    parser.add_argument("-W", "--workdir", action="store", dest="workdir", default="", help="working directory")
    return parser.parse_args(argv)


def _write_dist(path, rows):
    with open(path, "w") as outf:
        for t in rows:
            outf.write("\t".join(map(str, t)) + "\n")                     # :592-593: str() of every field


def main(argv=None):
    options = get_cmd(argv)
    print("Genome : {}".format(options.genome))
    base_genome = os.path.basename(options.genome)
    if not os.path.isdir(options.workdir):
        os.makedirs(options.workdir)
    if not options.conta:
        print("Contaminant : {}".format(None))
        output = os.path.join(options.workdir, base_genome + ".mcp_windows_vs_whole_" + options.dist + ".dist")
    else:
        base_conta = os.path.basename(options.conta)
        print("Contaminant : {} ".format(options.conta))
        output = base_genome + ".mcp_hostwindows_vs_"
        if options.host:
            print("Host : {}".format(options.host))
            output = os.path.join(options.workdir, output + "host_" + os.path.basename(options.host) + "_" + options.dist + ".dist")
        else:
            print("Host : None, using whole genome")
            output = os.path.join(options.workdir, output + "wholegenome_" + options.dist + ".dist")
    if not options.pattern and options.k:                                 # :552-555
        options.pattern = "1" * options.k
    # the reference always profiles the assembly itself as the first prototype (:558), also when -r is given
    genome = compute_whole_composition(options.genome, options.pattern, options.strand, nb_jobs=options.threads_max)
    if not options.conta:
        print("Computing microcomposition signaure and distances to genome")
    else:
        conta = compute_whole_composition(options.conta, options.pattern, options.strand, nb_jobs=options.threads_max)
    _write_dist(output, sliding_windows_distances(options.genome, genome, options.dist, options.pattern,
                                                  options.windows_size, options.windows_step, options))
    if options.conta:
        output = os.path.join(options.workdir, base_genome + ".mcp_hostwindows_vs_conta_" + base_conta + "_" + options.dist + ".dist")
        _write_dist(output, sliding_windows_distances(options.genome, conta, options.dist, options.pattern,
                                                      options.windows_size, options.windows_step, options))
    return 0


if __name__ == "__main__":
    main()
    sys.exit(0)

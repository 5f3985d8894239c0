"""CPU oracle for Kount.py's sliding-window scan (the ContaLocate front end).

TEST INFRASTRUCTURE ONLY -- same rules as phyloligo_oracle.py.  Parity status: PINNED by
tests/golden/kount.npz, produced by tests/golden/make_golden_kount.py from the reference's own
make_genome_chunk / compute_frequency / compute_whole_composition / compute_distances / main.

Reference lines restated (/root/reference/phylopackage/bin/Kount.py):
  :208-243  cut_sequence_and_count_pattern (strand selection inside, then as phyloligo.py)
  :245-272  count2freq
  :274-301  compute_frequency with the N gate (seq.count('N')/len(seq) <= n_max, else a NaN vector)
  :303-319  compute_whole_composition (counts summed over records, then one frequency vector)
  :69-123   KL / Eucl (x1000) / JSD (x1000)
  :322-330  compute_distance_joblib
  :343-407  make_genome_chunk: which windows exist and which coordinates are displayed
  :588-600  the .dist lines: id, displayed start, displayed stop, str(distance)
"""
from __future__ import annotations

import numpy as np

from . import phyloligo_oracle as po


def windows_of_record(length: int, wsize: int, wstep: int, min_windows_parallel: int = 20):
    """(start, displayed_start, displayed_stop) of every window of one record, in order.
    Three regimes (Kount.py:352-401): a record shorter than the window is one window (0, len);
    a record of fewer than 20 steps uses the 'serial' coordinate rules; longer records the
    'parallel' ones.  Windows start at range(0, len - wsize, wstep): a record exactly one window long
    yields nothing."""
    out = []
    if length < wsize:
        return [(0, 0, int(length))]
    starts = range(0, length - wsize, wstep)
    if length < min_windows_parallel * wstep:
        for s in starts:
            d0 = 1 if s == 0 else int(s + wsize / 2 - wstep / 2)
            d1 = length if s == length - wsize else int(s + wsize / 2 + wstep / 2)
            out.append((s, d0, d1))
    else:
        for s in starts:
            start, stop = int(s + wsize / 2 - wstep / 2), int(s + wsize / 2 + wstep / 2)
            d0 = 1 if start == (wsize / 2 - wstep / 2) else start
            edge = stop - wstep / 2 + wsize / 2
            d1 = length if (edge >= length - wstep and edge <= length) else stop
            out.append((s, d0, d1))
    return out


def record_id(title: str) -> str:
    parts = title.split(None, 1)
    return parts[0] if parts else ""


def whole_composition(seqs, pattern, strand) -> np.ndarray:
    counts, totals = po.compute_counts(seqs, pattern, strand)
    c = counts.sum(axis=0)
    t = int(totals.sum())
    return po.count2freq(c, t)


def window_gated(window: bytes, n_max: float) -> bool:
    """True when the window is NOT profiled: more than n_max of its bytes are upper-case 'N'."""
    return len(window) > 0 and (window.count(b"N") / len(window)) > float(n_max)


def _kl(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        d = a * np.log(a / b)
    d[~np.isfinite(d)] = 0
    return np.sum(d)


def distance(freq: np.ndarray, proto: np.ndarray, metric: str) -> float:
    if metric == "JSD":
        h = 0.5 * (freq + proto)
        return 0.5 * (_kl(freq, h) + _kl(proto, h)) * 1000
    if metric == "KL":
        return _kl(freq, proto)
    d = (freq - proto) ** 2
    d[~np.isfinite(d)] = 0
    return np.sqrt(np.sum(d)) * 1000


def scan(titles, seqs, proto, metric="JSD", pattern="1111", strand="both", wsize=5000, wstep=500, n_max=0.4):
    """[(id, displayed_start, displayed_stop, distance)] for every window of every record."""
    _, _, k, _ = po.pattern_info(pattern)
    rows = []
    for title, seq in zip(titles, seqs):
        for s, d0, d1 in windows_of_record(len(seq), wsize, wstep):
            w = seq[s:s + wsize]
            if window_gated(w, n_max):
                if k ** 4 != 4 ** k:
                    raise ValueError("operands could not be broadcast together (NaN vector of length k**4)")
                dist = 0.0 * (1000 if metric != "KL" else 1)       # NaN vector -> every term dropped -> 0
            else:
                dist = distance(po.compute_frequency(w, pattern, strand), proto, metric)
            rows.append((record_id(title), d0, d1, float(dist)))
    return rows


def dist_bytes(rows) -> bytes:
    return "".join("%s\t%s\t%s\t%s\n" % (i, a, b, str(np.float64(d))) for i, a, b, d in rows).encode()

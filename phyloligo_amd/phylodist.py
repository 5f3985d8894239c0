"""Per-pair metric functions with the names and call shape of the reference's metric library
(/root/reference/phylopackage/core/phylodist.py:36-85).  Each call runs the corresponding tile
kernel on a 2-row frequency matrix -- correct, but meant for interface parity and spot checks;
whole matrices go through phyloligo.compute_distances."""
import numpy as np

from .phyloligo import _context


def _pair(metric, a, b):
    f = np.ascontiguousarray(np.vstack([np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)]))
    return _context().pairwise_freq(f, metric)[0, 1]


def Eucl(a, b):
    return _pair("Eucl", a, b)


def JSD(a, b):
    return _pair("JSD", a, b)


def KT(a, b):
    return _pair("KT", a, b)


def BC(a, b):
    return _pair("BC", a, b)


def SC(a, b):
    return _pair("SC", a, b)

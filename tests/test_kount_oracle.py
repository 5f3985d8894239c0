"""Pin the Kount oracle (sliding windows vs prototype) against the reference's own outputs."""
import os

import numpy as np
import pytest

from oracle import kount_oracle as ko
from oracle import phyloligo_oracle as po


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "kount.npz"))


@pytest.mark.parametrize("key", ["1111_both", "11_plus", "1111_minus"])
def test_windows_gate_and_distances(gold, key):
    pattern, strand = key.split("_")
    titles, seqs = po.parse_fasta(gold["genome_fasta"].tobytes())
    proto = ko.whole_composition(seqs, pattern, strand)
    assert np.array_equal(proto, gold["proto_" + key])
    for metric in ("JSD", "KL", "Eucl"):
        rows = ko.scan(titles, seqs, proto, metric, pattern, strand, 1000, 200, 0.4)
        assert [r[0] for r in rows] == list(gold["win_id_" + key])
        assert [r[1] for r in rows] == list(gold["win_start_" + key])
        assert [r[2] for r in rows] == list(gold["win_stop_" + key])
        np.testing.assert_allclose([r[3] for r in rows], gold["dist_%s_%s" % (metric, key)], rtol=1e-13, atol=0)


def test_dist_files(gold):
    titles, seqs = po.parse_fasta(gold["genome_fasta"].tobytes())
    ctitles, cseqs = po.parse_fasta(gold["conta_fasta"].tobytes())
    proto = ko.whole_composition(seqs, "1111", "both")
    rows = ko.scan(titles, seqs, proto, "JSD", "1111", "both", 1000, 200, 0.4)
    want = gold["cli_whole_JSD__genome.fa.mcp_windows_vs_whole_JSD.dist"].tobytes()
    got = ko.dist_bytes(rows)
    assert got.count(b"\n") == want.count(b"\n")
    for g, w in zip(got.split(b"\n"), want.split(b"\n")):
        assert g.split(b"\t")[:3] == w.split(b"\t")[:3]
        if g:
            assert abs(float(g.split(b"\t")[3]) - float(w.split(b"\t")[3])) <= 1e-12 * abs(float(w.split(b"\t")[3]))
    proto2 = ko.whole_composition(seqs, "11", "plus")
    rows2 = ko.scan(titles, seqs, proto2, "Eucl", "11", "plus", 1000, 200, 0.4)
    want2 = gold["cli_whole_Eucl_k2__genome.fa.mcp_windows_vs_whole_Eucl.dist"].tobytes()
    assert len(rows2) == want2.count(b"\n")
    cproto = ko.whole_composition(cseqs, "1111", "both")
    rows3 = ko.scan(titles, seqs, cproto, "KL", "1111", "both", 1500, 300, 0.4)
    want3 = gold["cli_conta_KL__genome.fa.mcp_hostwindows_vs_conta_conta.fa_KL.dist"].tobytes().split(b"\n")
    assert len(rows3) == len(want3) - 1
    for r, w in zip(rows3, want3):
        f = w.split(b"\t")
        assert (r[0].encode(), str(r[1]).encode(), str(r[2]).encode()) == (f[0], f[1], f[2])
        assert abs(r[3] - float(f[3])) <= 1e-12 * max(1.0, abs(float(f[3])))


def test_window_rules():
    assert ko.windows_of_record(600, 1000, 200) == [(0, 0, 600)]
    assert ko.windows_of_record(1000, 1000, 200) == []                       # range(0, 0, step) is empty
    assert ko.windows_of_record(1200, 1000, 200) == [(0, 1, 600)]
    w = ko.windows_of_record(12345, 1000, 200)
    assert w[0] == (0, 1, 600) and w[1] == (200, 600, 800) and len(w) == len(range(0, 11345, 200))

"""Kount.py's sliding-window scan on the GPU against the reference's own outputs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "kount.npz"))


@pytest.fixture(scope="module")
def files(gold, tmp_path_factory):
    d = tmp_path_factory.mktemp("kount")
    g, c = d / "genome.fa", d / "conta.fa"
    g.write_bytes(gold["genome_fasta"].tobytes())
    c.write_bytes(gold["conta_fasta"].tobytes())
    return str(g), str(c), d


def _rows(path):
    out = []
    for ln in open(path, "rb").read().split(b"\n"):
        if ln:
            f = ln.split(b"\t")
            out.append((f[0], f[1], f[2], float(f[3])))
    return out


@pytest.mark.parametrize("key", ["1111_both", "11_plus", "1111_minus"])
def test_windows_and_distances_vs_reference(gold, files, key):
    from phyloligo_amd import kount
    pattern, strand = key.split("_")
    gpath = files[0]
    proto = kount.compute_whole_composition(gpath, pattern, strand)
    assert np.array_equal(proto, gold["proto_" + key])                 # exact counts -> bit-exact prototype
    opts = type("O", (), {"strand": strand, "n_max_freq_in_windows": 0.4})()
    for metric in ("JSD", "KL", "Eucl"):
        rows = kount.sliding_windows_distances(gpath, proto, metric, pattern, 1000, 200, opts)
        assert [r[0] for r in rows] == list(gold["win_id_" + key])
        assert [r[1] for r in rows] == list(gold["win_start_" + key])
        assert [r[2] for r in rows] == list(gold["win_stop_" + key])
        np.testing.assert_allclose([r[3] for r in rows], gold["dist_%s_%s" % (metric, key)], rtol=1e-6, atol=1e-12)


def test_cli_writes_reference_dist_files(gold, files):
    from phyloligo_amd import kount
    gpath, cpath, d = files
    runs = {"whole_JSD": (["-i", gpath, "-w", "1000", "-t", "200", "-d", "JSD", "-W", str(d / "o1")],
                          ["genome.fa.mcp_windows_vs_whole_JSD.dist"]),
            "whole_Eucl_k2": (["-i", gpath, "-w", "1000", "-t", "200", "-d", "Eucl", "-k", "2", "-s", "plus", "-W", str(d / "o2")],
                              ["genome.fa.mcp_windows_vs_whole_Eucl.dist"]),
            "conta_KL": (["-i", gpath, "-c", cpath, "-w", "1500", "-t", "300", "-d", "KL", "-W", str(d / "o3")],
                         ["genome.fa.mcp_hostwindows_vs_conta_conta.fa_KL.dist",
                          "genome.fa.mcp_hostwindows_vs_wholegenome_KL.dist"])}
    for name, (argv, outs) in runs.items():
        assert kount.main(argv) == 0
        for fn in outs:
            got = _rows(os.path.join(argv[-1], fn))
            want_raw = gold["cli_%s__%s" % (name, fn)].tobytes()
            want = [(f[0], f[1], f[2], float(f[3])) for f in (ln.split(b"\t") for ln in want_raw.split(b"\n") if ln)]
            assert len(got) == len(want) > 0
            for g, w in zip(got, want):
                assert g[:3] == w[:3]
                assert abs(g[3] - w[3]) <= 1e-6 * abs(w[3]) + 1e-12
            # gated windows are written as 0.0, like the reference
            assert sum(1 for g in got if g[3] == 0.0) == sum(1 for w in want if w[3] == 0.0)


def test_ranges_counting_matches_oracle(gold):
    """po_count_profiles_ranges on overlapping windows = the oracle's per-window counts, bit for bit."""
    import phyloligo_amd as pa
    from phyloligo_amd.phyloligo import _context
    from oracle import phyloligo_oracle as po
    titles, seqs = po.parse_fasta(gold["genome_fasta"].tobytes())
    seq = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    rng = np.random.default_rng(1)
    begins = rng.integers(0, len(seq) - 1, size=300)
    ends = np.minimum(len(seq), begins + rng.integers(0, 6000, size=300))
    for pattern, strand in (("1111", "both"), ("11011011", "minus"), ("101", "plus")):
        counts, totals = _context().count_profiles_ranges(seq, begins, ends, pattern, strand)
        for i in range(0, 300, 7):
            oc, ot = po.profile_counts(seq[begins[i]:ends[i]].tobytes(), pattern, strand)
            assert np.array_equal(counts[i].astype(np.int64), oc) and int(totals[i]) == ot

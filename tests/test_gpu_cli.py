"""End to end through the reference-shaped interface on the GPU: FASTA in, .mat out."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fasta(tmp_path_factory, golden_dir):
    g = np.load(os.path.join(golden_dir, "distances.npz"))
    contigs = [bytes(c) for c in g["contigs"]]
    path = tmp_path_factory.mktemp("cli") / "asm.fasta"
    with open(path, "wb") as fh:
        for i, s in enumerate(contigs):
            fh.write(b">contig_%d some description\n" % i)
            for p in range(0, len(s), 60):
                fh.write(s[p:p + 60] + b"\r\n")
    return str(path), g


@pytest.mark.parametrize("metric", ["Eucl", "JSD", "BC"])
def test_main_writes_reference_layout(fasta, tmp_path, metric, capsys):
    from phyloligo_amd import phyloligo as P
    path, g = fasta
    out = tmp_path / "out.mat"
    freq_out = tmp_path / "freq.tsv"
    rc = P.main(["-i", path, "-k", "4", "-s", "both", "-d", metric, "--method", "joblib", "-c", "8",
                 "-o", str(out), "-q", str(freq_out), "-w", str(tmp_path / "work")])
    assert rc == 0
    printed = capsys.readouterr().out.splitlines()
    assert printed == ["Using pattern 1111", "Computing frequencies", "Computing Pairwise distances",
                       "Writing frequency matrix", "Writing distance matrix"]        # phyloligo.py:1043-1065
    want = g["%s_1111_both" % metric]
    got = np.loadtxt(out, delimiter="\t")                                             # phyloselect.py:363
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-12, equal_nan=True)
    lines = out.read_bytes().split(b"\n")
    assert lines[-1] == b"" and len(lines) == want.shape[0] + 1
    for ln in lines[:-1]:
        fields = ln.split(b"\t")
        assert len(fields) == want.shape[0]
        assert all(f == b"nan" or (len(f) in (24, 25) and b"e" in f) for f in fields)   # "%.18e"
    assert np.array_equal(np.loadtxt(freq_out, delimiter="\t"), g["freq_1111_both"])
    if metric == "JSD":        # the reference's own .mat bytes: every field whose float64 value equals the
        # reference's must be the same bytes, and the file must parse back to exactly the values that were written
        ref_lines = g["matbytes_JSD_1111_both"].tobytes().split(b"\n")
        assert len(ref_lines) == len(lines)
        same = checked = 0
        for i, (l1, l2) in enumerate(zip(lines[:-1], ref_lines[:-1])):
            f1, f2 = l1.split(b"\t"), l2.split(b"\t")
            assert len(f1) == len(f2)
            for j, (a, b) in enumerate(zip(f1, f2)):
                if got[i, j] == want[i, j] or (np.isnan(got[i, j]) and np.isnan(want[i, j])):
                    checked += 1
                    assert a == b, (i, j, a, b)
                    same += 1
        assert checked >= want.shape[0]          # at least the diagonal (exact zeros) is value-identical
        # and wherever the value differs in the last bits the field is still the correct "%.18e" of OUR value
        import io
        buf = io.BytesIO()
        np.savetxt(buf, got, delimiter="\t")
        assert buf.getvalue() == out.read_bytes()


def test_memmap_container(fasta, tmp_path):
    from phyloligo_amd import phyloligo as P
    path, g = fasta
    out = tmp_path / "out.f32"
    P.main(["-i", path, "-p", "1111", "-d", "JSD", "--method", "joblib", "--large", "memmap", "-o", str(out)])
    raw = np.fromfile(out, dtype=np.float32)                 # phyloligo_comparemat.py:16-24: N = sqrt(len)
    n = int(round(np.sqrt(raw.shape[0])))
    assert n * n == raw.shape[0] == g["JSD_1111_both"].size
    np.testing.assert_allclose(raw.reshape(n, n), g["JSD_1111_both"], rtol=1e-6, atol=1e-7)
    assert np.allclose(raw.reshape(n, n), g["JSD_1111_both"], atol=1e-3)      # the reference's own criterion


@pytest.mark.parametrize("ranks", [1, 2])
def test_memmap_container_over_an_existing_file(fasta, tmp_path, ranks):
    """An output path that already holds a file - longer, shorter, or of exactly the new size - is reused without being emptied
    first (O_TRUNC on a cached 10 GB container costs as much as writing it): the result must be the same bytes and the same
    length as into a new file, from one process and from two ranks."""
    import subprocess
    import sys
    from phyloligo_amd import phyloligo as P
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path, g = fasta
    fresh = tmp_path / "fresh.f32"
    args = ["-i", path, "-p", "1111", "-d", "BC", "--method", "joblib", "--large", "memmap"]

    def run(out):
        if ranks == 1:
            P.main(args + ["-o", str(out)])
        else:
            env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
            env.update(PO_CLI_REHEARSAL="1", MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
            r = subprocess.run([sys.executable, "-m", "phyloligo_amd", "--gpus", "2"] + args + ["-o", str(out)], capture_output=True,
                               text=True, timeout=600, cwd=ROOT, env=env)
            assert r.returncode == 0, r.stderr[-2000:]

    run(fresh)
    want = fresh.read_bytes()
    assert len(want) == g["BC_1111_both"].size * 4
    for name, old in (("longer", b"\xff" * (len(want) + 12345)), ("shorter", b"\xff" * 1000), ("same", b"\xff" * len(want))):
        out = tmp_path / (name + ".f32")
        out.write_bytes(old)
        run(out)
        assert out.read_bytes() == want, name


def _read_memmap_like_comparemat(path):
    """phyloligo_comparemat.py:16-24 restated: float32 memmap, N = sqrt(len), weird shapes rejected"""
    matrix = np.memmap(path, dtype=np.float32, mode="r")
    n = np.sqrt(matrix.shape[0])
    assert str(n).split(".")[1] == "0", "weird shape"
    return np.array(matrix.reshape((int(n), int(n))))


@pytest.mark.parametrize("pattern,strand", [("1111", "both"), ("1101", "minus")])
@pytest.mark.parametrize("metric", ["Eucl", "JSD", "BC"])
def test_memmap_container_vs_the_references_own_memmap_bytes(fasta, golden_dir, tmp_path, metric, pattern, strand, capsys):
    """SURVEY 8f-1 pinned by the reference itself: tests/golden/memmap.npz holds the BYTES of the
    container that the reference's compute_distances_memmap (bin/phyloligo.py:394-427, euclidean_distances_loc / JSD_loc /
    BC_loc :200-217) wrote from a float32 frequency memmap filled by its compute_frequency_memmap (:693-720).  The HIP
    container, read back exactly as phyloligo_comparemat.py:16-24 reads one, must pass the reference's criterion
    (:44, numpy.allclose(atol=1e-3)) against those bytes; the measured deviation is printed (the reference computes this
    variant in float32, the HIP path in float64 with one rounding on store)."""
    from phyloligo_amd import phyloligo as P
    path, g = fasta
    mm = np.load(os.path.join(golden_dir, "memmap.npz"))
    assert [bytes(c) for c in mm["contigs"]] == [bytes(c) for c in g["contigs"]]
    out = tmp_path / "out.f32"
    P.main(["-i", path, "-p", pattern, "-s", strand, "-d", metric, "--method", "joblib", "--large", "memmap", "-o", str(out)])
    capsys.readouterr()
    got = _read_memmap_like_comparemat(out)
    ref_file = tmp_path / "ref.f32"
    mm["container_%s_%s_%s" % (metric, pattern, strand)].tofile(ref_file)
    want = _read_memmap_like_comparemat(ref_file)
    assert got.shape == want.shape == (48, 48)
    assert np.array_equal(np.isnan(got), np.isnan(want))                      # BC of two empty profiles: nan in both
    assert np.allclose(got, want, atol=1e-3, equal_nan=True)                  # phyloligo_comparemat.py:44 (+ the nan pattern above)
    fin = ~np.isnan(want)
    dev = float(np.max(np.abs(got[fin].astype(np.float64) - want[fin].astype(np.float64))))
    with capsys.disabled():
        print("\n[memmap %s %s %s] max |HIP container - reference container| = %.3g (largest entry %.3g)"
              % (metric, pattern, strand, dev, float(np.max(np.abs(want[fin])))))
    # far inside the criterion: float32 arithmetic of the reference (sklearn's float32 Gram form for Eucl) vs one rounding here
    assert dev < (2e-4 if metric == "Eucl" else 2e-6)
    # the frequency container of that variant is float32 too (:905): the HIP frequencies rounded once are those values
    freq, _ = P.compute_frequencies("joblib", "memmap", path, pattern, strand, 250, 4, str(tmp_path))
    assert np.array_equal(np.asarray(freq).astype(np.float32), mm["freq32_%s_%s" % (pattern, strand)])


def test_reference_memmap_variant_known_failures(golden_dir):
    """What the reference's memmap variant cannot do, as recorded when the fixture was made: SC raises NameError (spearmanr
    is never imported into phylodist, core/phylodist.py:82-85); KT needs Bio.Cluster, which is absent from this image (the
    recorded error is the stand-in module's, not the reference's) - neither has reference bytes to compare with."""
    mm = np.load(os.path.join(golden_dir, "memmap.npz"))
    assert "NameError" in str(mm["raises_SC_1111_both"][0]) and "spearmanr" in str(mm["raises_SC_1111_both"][0])
    assert "Bio.Cluster" in str(mm["raises_KT_1111_both"][0])
    assert "container_KT_1111_both" not in mm.files and "container_SC_1111_both" not in mm.files


def test_dispatcher_functions_and_errors(fasta, capsys):
    from phyloligo_amd import phyloligo as P, phylodist
    path, g = fasta
    freq, name = P.compute_frequencies("joblib", "None", path, "11", "plus", 250, 4, ".")
    assert name is None and np.array_equal(np.asarray(freq), g["freq_11_plus"])
    res = P.compute_distances("joblib", "None", freq, None, "unused", "Eucl", 4, 250, ".")
    np.testing.assert_allclose(res, g["Eucl_11_plus"], rtol=1e-6, atol=1e-12)
    # a plain ndarray of frequencies (what the reference passes) takes the frequency entry point
    res2 = P.compute_distances("joblib", "None", np.array(freq), None, "unused", "JSD", 4, 250, ".")
    np.testing.assert_allclose(res2, g["JSD_11_plus"], rtol=1e-6, atol=1e-12)
    with pytest.raises(SystemExit) as e:
        P.compute_distances("joblib", "None", freq, None, "unused", "XX", 4, 250, ".")
    assert e.value.code == 1
    with pytest.raises(SystemExit) as e:
        P.compute_frequencies("joblib", "None", path, "11", "sideways", 250, 4, ".")
    assert e.value.code == 1
    assert P.compute_distances("mpi", "None", freq, None, "unused", "Eucl", 4, 250, ".") is None
    a, b = g["freq_1111_both"][0], g["freq_1111_both"][1]
    np.testing.assert_allclose(phylodist.JSD(a, b), g["JSD_1111_both"][0, 1], rtol=1e-6)
    np.testing.assert_allclose(phylodist.Eucl(a, b), g["Eucl_1111_both"][0, 1], rtol=1e-6)
    np.testing.assert_allclose(phylodist.BC(a, b), g["BC_1111_both"][0, 1], rtol=1e-6)


def test_edited_profile_matrix_goes_by_its_values(fasta):
    """in-place edits of the returned frequencies must not be ignored in favour of the integer profiles
    that ride on the ProfileMatrix (the reference computes from the array it is handed, phyloligo.py:536-553)."""
    from phyloligo_amd import phyloligo as P
    from oracle import phyloligo_oracle as po
    path, g = fasta
    freq, _ = P.compute_frequencies("joblib", "None", path, "1111", "both", 250, 4, ".")
    assert freq.counts is not None
    freq[3, :] = 0.0                      # in place: attributes survive
    freq[5, :17] *= 0.5
    assert freq.counts is not None
    res = P.compute_distances("joblib", "None", freq, None, "unused", "Eucl", 4, 250, ".")
    want = po.pairwise_block(np.array(freq), "Eucl")
    np.testing.assert_allclose(res, want, rtol=1e-6, atol=1e-12)
    assert abs(res[3, 0] - g["Eucl_1111_both"][3, 0]) > 1e-6        # i.e. NOT the stale profile's distance


def test_scoop_ignores_large_and_writes_text(fasta, tmp_path):
    """compute_distances_scoop has no memmap variant (phyloligo.py:313-362); main writes the text matrix
    for scoop whatever --large says (:1064)."""
    from phyloligo_amd import phyloligo as P
    path, g = fasta
    out = tmp_path / "scoop.mat"
    assert P.main(["-i", path, "-k", "4", "-d", "Eucl", "--method", "scoop", "--large", "memmap", "-o", str(out)]) == 0
    got = np.loadtxt(out, delimiter="\t")
    np.testing.assert_allclose(got, g["Eucl_1111_both"], rtol=1e-6, atol=1e-12)


def test_out_buffer_is_validated():
    import phyloligo_amd as pa
    from phyloligo_amd._lib import PhyloligoError
    rng = np.random.default_rng(0)
    counts = rng.integers(0, 20, size=(40, 16)).astype(np.uint32)
    totals = counts.sum(axis=1).astype(np.uint64)
    with pa.Context(0) as ctx:
        for bad in (np.zeros((40, 40), np.float32), np.zeros((40, 39)), np.asfortranarray(np.zeros((40, 40))),
                    np.zeros((40, 80))[:, ::2]):
            with pytest.raises(PhyloligoError):
                ctx.pairwise(counts, totals, "Eucl", out=bad)
        with pytest.raises(PhyloligoError):
            pa.api.write_mat_text("/tmp/never_written.mat", None)
        ok = np.full((40, 48), -1.0)
        ctx.pairwise(counts, totals, "Eucl", out=ok)
        assert (ok[:, 40:] == -1.0).all() and (ok[:, :40] >= 0).all()


def test_row_block_paths_of_the_dispatcher(fasta, tmp_path, monkeypatch):
    """Matrices above 4 GB are computed and copied in row blocks, and the raw float32 container is written block by
    block from two host buffers: force both paths at a small size and compare with the one-call result."""
    from phyloligo_amd import phyloligo as P
    path, g = fasta
    freq, _ = P.compute_frequencies("joblib", "None", path, "1111", "both", 250, 4, ".")
    whole = P.compute_distances("joblib", "None", freq, None, "unused", "JSD", 4, 250, ".")
    monkeypatch.setattr(P, "_SINGLE_CALL_BYTES", 1024)
    monkeypatch.setattr(P, "_row_chunk", lambda n, itemsize, budget=0: 7)          # 7-row blocks: ragged last block
    blocks = P.compute_distances("joblib", "None", freq, None, "unused", "JSD", 4, 250, ".")
    np.testing.assert_allclose(blocks, whole, rtol=1e-12, atol=1e-15)
    out = tmp_path / "blocks.f32"
    assert P.compute_distances("joblib", "memmap", freq, None, str(out), "JSD", 4, 250, ".") is None
    raw = np.fromfile(out, dtype=np.float32).reshape(whole.shape)
    assert np.array_equal(raw, whole.astype(np.float32)) or np.allclose(raw, whole, rtol=1e-6, atol=1e-7)


def test_bc_from_frequencies_takes_the_thermometer_path():
    """The reference's dispatcher hands over float64 frequencies: count2freq output is traced back to the integer
    profiles on the device and then takes the same kernels as the count entry point (here: BC on thermometer planes)."""
    import phyloligo_amd as pa
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    contigs = [acgt[rng.integers(0, 4, size=1500)].tobytes() for _ in range(300)]
    seq = np.frombuffer(b"".join(contigs), dtype=np.uint8).copy()
    off = np.arange(301, dtype=np.uint64) * np.uint64(1500)
    with pa.Context(0) as ctx:
        counts, totals = ctx.count_profiles(seq, off, "11011011", "both")
        a, st_a = ctx.pairwise(counts, totals, "BC", want_stats=True)
        b, st_b = ctx.pairwise_freq(ctx.frequencies(counts, totals), "BC", want_stats=True)
        assert st_a["kernel_id"] == 9 and st_b["kernel_id"] == 9
        assert np.array_equal(a, b)


@pytest.mark.parametrize("ranks", [2, 3])
def test_multi_gpu_cli_rehearsal(tmp_path, ranks):
    """python -m torch.distributed.run -m phyloligo_amd: one process per GPU, every rank profiles its contigs, one
    all-gather, tournament row blocks, row-completing exchange, every rank writes its rows.  Here the ranks share this
    one GPU over gloo (PO_CLI_REHEARSAL=1); the files must equal the single-process run byte for byte - text matrix,
    frequency matrix and the float32 container (ragged lengths, a block boundary that is not a multiple of anything)."""
    import subprocess
    import sys
    from phyloligo_amd import phyloligo as P
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(77)
    fa = tmp_path / "asm.fa"
    with open(fa, "wb") as fh:
        for i in range(517):
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(300, 5000)))].tobytes()
            fh.write(b">c%d\n" % i + s + b"\n")
    env = dict(os.environ, PO_CLI_REHEARSAL="1", MASTER_ADDR="127.0.0.1", PYTHONPATH=root)
    for metric, large, name in (("JSD", "None", "jsd.mat"), ("KT", "None", "kt.mat"), ("Eucl", "memmap", "eucl.f32")):
        ref, got = tmp_path / ("ref_" + name), tmp_path / ("got_" + name)
        args = ["-i", str(fa), "-k", "4", "-d", metric, "--method", "joblib", "--large", large]
        assert P.main(args + ["-o", str(ref), "-q", str(ref) + ".freq"]) == 0
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                              "--master-addr", "127.0.0.1", "--master-port", str(29540 + ranks), "-m", "phyloligo_amd"] + args +
                             ["-o", str(got), "-q", str(got) + ".freq"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        assert out.stdout.count("Computing Pairwise distances") == 1            # rank 0 speaks
        assert open(ref, "rb").read() == open(got, "rb").read(), (metric, large)
        assert open(str(ref) + ".freq", "rb").read() == open(str(got) + ".freq", "rb").read()
        if metric == "Eucl":
            # the same job started plainly: `python -m phyloligo_amd --gpus N ...` starts its own ranks (phyloligo_amd/launch.py)
            env2 = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
            got2 = tmp_path / ("plain_" + name)
            out = subprocess.run([sys.executable, "-m", "phyloligo_amd", "--gpus", str(ranks)] + args + ["-o", str(got2)],
                                 capture_output=True, text=True, timeout=600, cwd=root, env=env2)
            assert out.returncode == 0, out.stderr[-3000:]
            assert out.stdout.count("Computing Pairwise distances") == 1
            assert open(ref, "rb").read() == open(got2, "rb").read()


def test_cli_process_runs_without_torch(fasta, tmp_path):
    """`python -m phyloligo_amd` as one process needs no torch (numpy + the host-pointer entry points of the C ABI): same
    bytes as the in-process run, and torch is never imported (0.7 s of a 2 - 3 s run at 50 000 contigs)."""
    import subprocess
    import sys
    from phyloligo_amd import phyloligo as P
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path, g = fasta
    ref, got = tmp_path / "ref.mat", tmp_path / "got.mat"
    args = ["-i", path, "-k", "4", "-d", "JSD", "--method", "joblib"]
    assert P.main(args + ["-o", str(ref)]) == 0
    code = ("import sys, runpy\n"
            "sys.argv = ['phyloligo_amd'] + %r\n"
            "try:\n    runpy.run_module('phyloligo_amd', run_name='__main__')\n"
            "except SystemExit as e:\n    assert e.code in (0, None)\n"
            "print('TORCH_IMPORTED' if 'torch' in sys.modules else 'NO_TORCH')\n") % (args + ["-o", str(got)],)
    env = dict(os.environ, PYTHONPATH=root)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "NO_TORCH" in out.stdout and "TORCH_IMPORTED" not in out.stdout
    assert open(ref, "rb").read() == open(got, "rb").read()
    raw_ref, raw_got = tmp_path / "ref.f32", tmp_path / "got.f32"
    assert P.main(args + ["--large", "memmap", "-o", str(raw_ref)]) == 0
    out = subprocess.run([sys.executable, "-m", "phyloligo_amd"] + args + ["--large", "memmap", "-o", str(raw_got)],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert open(raw_ref, "rb").read() == open(raw_got, "rb").read()


def test_json_stats(fasta, tmp_path, capsys):
    """--json-stats: sizes, phase times, library and device, the stage-2 kernel of the run - next to the
    five progress lines, which stay exactly the reference's."""
    import json
    from phyloligo_amd import phyloligo as P
    path, g = fasta
    out, js = tmp_path / "o.mat", tmp_path / "stats.json"
    assert P.main(["-i", path, "-k", "4", "-d", "JSD", "--method", "joblib", "-o", str(out), "--json-stats", str(js)]) == 0
    assert capsys.readouterr().out.splitlines() == ["Using pattern 1111", "Computing frequencies", "Computing Pairwise distances",
                                                    "Writing distance matrix"]
    st = json.loads(js.read_text())
    assert st["contigs"] == 48 and st["words"] == 256 and st["pairs"] == 48 * 47 // 2 and st["metric"] == "JSD" and st["gpus"] == 1
    assert st["library"].startswith("phyloligo_amd") and len(st["device"]) > 0
    assert set(st["seconds"]) == {"frequencies", "distances", "writing", "total"} and st["seconds"]["total"] > 0
    assert st["stage2_first_call"]["kernel_id"] in (1, 6) and st["stage2_first_call"]["rows"] == [0, 48]
    # round 5: every step of the ingest with its own time (device parser when torch is loaded, host parser otherwise)
    ph = st["ingest_phases_ms"]
    assert ("file_read_ms" in ph and "stage1_ms" in ph and "frequencies_d2h_ms" in ph) or ("host_parse_ms" in ph and "stage1_host_pointers_ms" in ph)
    assert all(v >= 0 for v in ph.values()) and sum(ph.values()) <= st["seconds"]["frequencies"] * 1e3 * 1.05 + 1.0
    np.testing.assert_allclose(np.loadtxt(out, delimiter="\t"), g["JSD_1111_both"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("metric", ["Eucl", "JSD", "BC"])
def test_h5py_container_equals_the_memmap_container(fasta, tmp_path, metric, capsys):
    """--large h5py (bin/phyloligo.py:456-534): one HDF5 file with the float32 dataset "distances" (N, N), as
    join_distance_results writes it and phyloselect.py:615-619 reads it.  Same mathematics as the memmap variant
    (euclidean_distances_h5py / JSD_h5py / BC_h5py mirror the *_loc functions), so libhdf5 must read back exactly the float32
    values of the raw container - which is pinned by the reference's own bytes (test_memmap_container_vs_...)."""
    from phyloligo_amd import hdf5, phyloligo as P
    if not hdf5.available():
        pytest.skip("no libhdf5 >= 1.10 on this system")
    path, g = fasta
    raw, h5 = tmp_path / "out.f32", tmp_path / "out.h5"
    args = ["-i", path, "-p", "1111", "-d", metric, "--method", "joblib"]
    assert P.main(args + ["--large", "memmap", "-o", str(raw)]) == 0
    assert P.main(args + ["--large", "h5py", "-o", str(h5)]) == 0
    printed = capsys.readouterr().out.splitlines()
    assert printed[-3:] == ["Using pattern 1111", "Computing frequencies", "Computing Pairwise distances"]    # no text matrix (:1064)
    got = hdf5.read_f32_dataset(str(h5), "distances")
    want = np.fromfile(raw, dtype=np.float32).reshape(48, 48)
    assert got.shape == (48, 48) and np.array_equal(got, want, equal_nan=True)
    # h5py itself, where an interpreter of this system has it (the image's conda python3.9: h5py 3.3.0): the reference's read
    import subprocess
    exe = "/opt/conda/bin/python3.9"
    if os.path.exists(exe) and subprocess.run([exe, "-c", "import h5py"], capture_output=True).returncode == 0:
        code = ("import h5py, numpy, sys\n"
                "with h5py.File(sys.argv[1], 'r') as hf:\n"
                "    matrix = hf.get('distances'); matrix = matrix[...]\n"          # phyloselect.py:617-619 (.value is h5py < 3)
                "numpy.save(sys.argv[2], matrix)\n")
        r = subprocess.run([exe, "-c", code, str(h5), str(tmp_path / "via_h5py.npy")], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        assert np.array_equal(np.load(tmp_path / "via_h5py.npy"), want, equal_nan=True)


def test_h5py_container_from_two_ranks(tmp_path):
    """the same HDF5 container written by two ranks (rehearsal on one GPU): rank 0 lets libhdf5 create the file, every rank fills
    its rows of the dataset's data range; libhdf5 reads back the single-process values"""
    import subprocess
    import sys
    from phyloligo_amd import hdf5, phyloligo as P
    if not hdf5.available():
        pytest.skip("no libhdf5 >= 1.10 on this system")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(3)
    fa = tmp_path / "asm.fa"
    with open(fa, "wb") as fh:
        for i in range(301):
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(300, 3000)))].tobytes()
            fh.write(b">c%d\n" % i + s + b"\n")
    one, two = tmp_path / "one.h5", tmp_path / "two.h5"
    args = ["-i", str(fa), "-k", "4", "-d", "JSD", "--method", "joblib", "--large", "h5py"]
    assert P.main(args + ["-o", str(one)]) == 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PO_CLI_REHEARSAL="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-m", "phyloligo_amd", "--gpus", "2"] + args + ["-o", str(two)],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    a, b = hdf5.read_f32_dataset(str(one), "distances"), hdf5.read_f32_dataset(str(two), "distances")
    assert a.shape == (301, 301) and np.array_equal(a, b)

"""CPU-side checks of the product: the C-ABI library loads and exports what include/*.h
declares, host formats (FASTA ingest, .mat text) match the reference's, CLI resolution matches
the reference's argparse, and compute calls fail loudly without a GPU.  No compute on CPU."""
import io
import os
import re
import sys

import numpy as np
import pytest

import phyloligo_amd as pa
from phyloligo_amd import _lib, phyloligo as P
from oracle import phyloligo_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "phyloligo_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(po_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libphyloligo_amd.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), "ctypes table and header out of step"
    assert lib.po_abi_version() == 1
    assert b"gfx950" in lib.po_version()


def test_pattern_info_and_errors():
    assert pa.pattern_info("1111") == (4, 4, 256)
    assert pa.pattern_info(4) == (4, 4, 256)
    assert pa.pattern_info("11011011") == (8, 6, 4096)
    assert pa.pattern_info("1" + "0" * 30 + "1") == (32, 2, 16)
    assert pa.pattern_info("1" + "0" * 31 + "1") == (33, 2, 16)            # wider than 32: two 64-bit window halves
    assert pa.pattern_info("11" + "0" * 30 + "101" + "0" * 27 + "11") == (64, 6, 4096)
    for bad in ("", "000", "12", "1" * 9, "1" + "0" * 63 + "1"):
        with pytest.raises(pa.PhyloligoError):
            pa.pattern_info(bad)
    with pytest.raises(pa.PhyloligoError) as e:
        pa.pattern_info("1" + "0" * 63 + "1")                                 # 65 positions: outside the envelope
    assert e.value.status == _lib.PO_EUNSUPPORTED


def test_no_cpu_fallback():
    if pa.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pa.PhyloligoError) as e:
        pa.Context(0)
    assert e.value.status == _lib.PO_ENODEV
    with pytest.raises(pa.PhyloligoError):
        P.compute_frequencies("joblib", "None", __file__, "1111", "both")


FASTA_CASES = [
    b">a desc\nACGT\nAC GT\r\n\n>b\n>c\nNN\nacgt  \n",
    b"\n\n>only\nACGTACGT",
    b">x\n\n\nAC\tGT \t\nTT\n>y  trailing  \nGG\x0b\n",
    b">e1\n>e2\n>e3\n",
    b"",
    b">last\nAC\r\nGT\r\n",
]


from tests.fasta_cases import CASES as SHARED_FASTA_CASES

FASTA_CASES = FASTA_CASES + [SHARED_FASTA_CASES[k] for k in sorted(SHARED_FASTA_CASES)]


@pytest.mark.parametrize("data", FASTA_CASES)
def test_fasta_index_matches_biopython_semantics(data):
    titles, seqs = po.parse_fasta(data)
    seq, offsets, got_titles = pa.fasta_index(data)
    assert got_titles == titles
    assert len(offsets) == len(seqs) + 1
    for i, s in enumerate(seqs):
        assert seq[int(offsets[i]):int(offsets[i + 1])].tobytes() == s


def test_fasta_rejects_leading_text():
    with pytest.raises(pa.PhyloligoError):
        pa.fasta_index(b"ACGT\n>a\nAC\n")


def test_fasta_wrapped_synthetic_roundtrip():
    from phyloligo_amd import synthetic
    seq, offsets = synthetic.contig_bytes(37, 333, seed=5)
    data = synthetic.fasta_bytes(seq, offsets)
    seq2, off2, titles = pa.fasta_index(data)
    assert np.array_equal(seq, seq2) and np.array_equal(offsets, off2)
    assert titles[0] == "c0000000" and titles[-1] == "c0000036"
    assert data == po.fasta_bytes(po.synthetic_contigs(37, 333, seed=5))


def test_mat_text_is_numpy_savetxt(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "distances.npz"))
    m = g["JSD_1111_both"]
    path = tmp_path / "a.mat"
    pa.write_mat_text(path, m)
    assert path.read_bytes() == g["matbytes_JSD_1111_both"].tobytes()     # the reference's own bytes
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(500) * 10.0 ** rng.integers(-300, 300, 500),
                        [0.0, -0.0, np.nan, np.inf, -np.inf, 1e-310, 5e-324, 1.7976931348623157e308]]).reshape(-1, 4)
    pa.write_mat_text(path, x)
    buf = io.BytesIO()
    np.savetxt(buf, x, delimiter="\t")
    assert path.read_bytes() == buf.getvalue()
    pa.write_mat_text(path, x[:2], append=True)
    assert path.read_bytes() == buf.getvalue() + buf.getvalue()[:len(buf.getvalue().split(b"\n")[0]) * 0 + sum(len(l) + 1 for l in buf.getvalue().split(b"\n")[:2])]
    # an existing file is overwritten in place (no O_TRUNC: that costs as much as the write on a large cached file) and cut
    # to the new length; a matrix large enough for several rounds of slabs on several threads, written over a longer file
    big = rng.standard_normal((700, 900)) * 10.0 ** rng.integers(-5, 5, (700, 900))
    path.write_bytes(b"x" * 30_000_000)
    pa.write_mat_text(path, big)
    buf = io.BytesIO()
    np.savetxt(buf, big, delimiter="\t")
    assert path.read_bytes() == buf.getvalue()
    pa.write_mat_text(path, x)                                          # and a short one over the long one
    buf = io.BytesIO()
    np.savetxt(buf, x, delimiter="\t")
    assert path.read_bytes() == buf.getvalue()


def test_mat_text_into_a_pipe_and_dev_null(tmp_path):
    """Not every -o is a regular file: a FIFO (what `-o /dev/stdout | ...` is) gets the rows in order through write(), /dev/null
    is not truncated (ftruncate fails there) - both worked with numpy.savetxt and keep working."""
    import threading
    rng = np.random.default_rng(3)
    m = rng.standard_normal((400, 300))
    fifo = str(tmp_path / "pipe")
    os.mkfifo(fifo)
    got = []
    reader = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    reader.start()
    pa.write_mat_text(fifo, m)
    reader.join(30)
    buf = io.BytesIO()
    np.savetxt(buf, m, delimiter="\t")
    assert got and got[0] == buf.getvalue()
    pa.write_mat_text("/dev/null", m)


def test_cli_resolution_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "cli.npz"))
    for k in z.files:
        argv = str(z[k][0]).split()
        p = P.get_cmd(argv)
        pat = p.pattern
        if type(pat) == int:
            pat = "1" * pat
        assert [pat, p.strand, p.dist, p.out_file, p.large, str(p.threads_max)] == [str(x) for x in z[k][1:]], k
    with pytest.raises(SystemExit) as e:          # --method is required, argparse exits 2
        P.get_cmd(["-i", "x.fa"])
    assert e.value.code == 2
    with pytest.raises(SystemExit):
        P.get_cmd(["-i", "x.fa", "--method", "joblib", "-d", "XX"])


def test_synthetic_generators_agree():
    from phyloligo_amd import synthetic
    seq, offsets = synthetic.contig_bytes(20, 100, seed=5)
    part, _ = synthetic.contig_bytes_range(20, 100, 5, 7, 13)
    assert np.array_equal(seq[700:1300], part)
    assert b"".join(po.synthetic_contigs(20, 100, seed=5)) == seq.tobytes()


def test_kount_window_rules_match_reference(golden_dir):
    """Host logic of the Kount mirror: which windows exist and the coordinates that are displayed."""
    from phyloligo_amd import kount
    g = np.load(os.path.join(golden_dir, "kount.npz"))
    titles, seqs = po.parse_fasta(g["genome_fasta"].tobytes())
    ids, d0, d1, lens = [], [], [], []
    for t, s in zip(titles, seqs):
        for start, a, b in kount.record_windows(len(s), 1000, 200):
            ids.append(t.split(None, 1)[0]); d0.append(a); d1.append(b); lens.append(min(len(s), start + 1000) - start)
    assert ids == list(g["win_id_1111_both"]) and d0 == list(g["win_start_1111_both"]) and d1 == list(g["win_stop_1111_both"])
    assert lens == list(g["win_len_1111_both"])
    p = kount.get_cmd(["-i", "x.fa"])
    assert (p.k, p.pattern, p.windows_size, p.windows_step, p.dist, p.strand, p.n_max_freq_in_windows) == \
        (4, None, 5000, 500, "JSD", "both", 0.4)


def test_fasta_parallel_segments_equal_sequential_semantics():
    """A file large enough to be cut into several host-thread segments (>= 4 MiB each): ragged line lengths, CRLF,
    blank lines, spaces, '>' inside sequence lines, empty records - against the oracle's line-by-line parser."""
    rng = np.random.default_rng(99)
    alphabet = np.frombuffer(b"ACGTacgtNn> \t", dtype=np.uint8)
    probs = np.array([22, 22, 22, 22, 2, 2, 2, 2, 1, 1, 0.5, 1, 0.5])
    probs = probs / probs.sum()
    parts = [b"\n  \n"]                                            # blank prelude
    for r in range(6000):
        parts.append(b">rec%d some description %d \r\n" % (r, r * 7))
        for _ in range(int(rng.integers(0, 60))):
            line = alphabet[rng.choice(len(alphabet), size=int(rng.integers(0, 120)), p=probs)].tobytes()
            if line.startswith(b">"):
                line = b"A" + line
            parts.append(line + (b"\r\n" if r % 3 == 0 else b"\n"))
        if r % 500 == 0:
            parts.append(b"\n\n")
    data = b"".join(parts)
    assert len(data) > 10 << 20                                   # >= 3 segments of >= 4 MiB
    titles, seqs = po.parse_fasta(data)
    seq, offsets, got_titles = pa.fasta_index(data)
    assert len(got_titles) == len(titles) == 6000
    assert got_titles[0] == titles[0] and got_titles[5999] == titles[5999] and got_titles[1234:1237] == titles[1234:1237]
    assert int(offsets[-1]) == sum(len(s) for s in seqs) == seq.shape[0]
    assert seq.tobytes() == b"".join(seqs)
    assert np.array_equal(np.diff(offsets.astype(np.int64)), np.array([len(s) for s in seqs]))


def test_container_is_sized_and_zero_before_the_rows_arrive(tmp_path):
    """_reserve_file: the float32 container has its final size (blocks allocated where the filesystem can) and reads as
    zeros, exactly like the truncated file it replaces (reference: numpy.memmap(mode='w+'), phyloligo.py:413)."""
    path = str(tmp_path / "m.f32")
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o666)
    try:
        P._reserve_file(fd, 3 * 1000 * 4)
        P._reserve_file(fd, 0)                       # n = 0: an empty container, no error
    finally:
        os.close(fd)
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o666)
    try:
        P._reserve_file(fd, 1 << 20)
    finally:
        os.close(fd)
    assert os.path.getsize(path) == 1 << 20
    assert not np.fromfile(path, dtype=np.uint8).any()


def test_scalar_row_loads_are_waited_for_before_use():
    """jsd_lut_rows_kernel issues s_load_dwordx16 in one asm statement and waits in a later one; the gfx950
    assembly of the file as built must not touch the destination SGPRs in between (tools/check_scalar_loads.py)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scalar_loads.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout


def test_no_kernel_spills_registers_to_scratch():
    """Scratch memory is HBM traffic no algorithmic byte count knows about (3.9 GB of extra writes per
    matrix came from 27 spilled registers of valu_tile_kernel<JSD>).  Every kernel inside the BUILT library (the gfx950 code
    objects of its .hip_fatbin section, metadata read with llvm-readelf) must have no private segment - except the one on the
    allow list of tools/check_spills.py (`--compile` checks the assembly of a fresh compilation instead)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_spills.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "outside the allow list: 0" in r.stdout


def test_library_and_committed_counters_belong_to_these_sources():
    """po_version() carries the hash of the sources the library was built from (tools/source_hash.py); the counter files
    bench.py quotes (profiles/traffic.json, pmc_busy.json) carry the hash of the library they were measured on.  All three
    must be the tree's: a kernel edit without `bash tools/profile_round.sh` on the GPU box fails here, before it can put stale
    counters into a driver-run record (which happened in round 2)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import source_hash
    tree = source_hash()
    lib = _lib.load().po_version().decode()
    assert lib.endswith("src " + tree), "libphyloligo_amd.so is not built from these sources: %s vs tree %s" % (lib, tree)
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    busy = json.load(open(os.path.join(ROOT, "profiles", "pmc_busy.json")))
    assert traffic["_detail"].get("src_hash") == tree, "profiles/traffic.json was measured on another build: re-run tools/profile_round.sh"
    assert busy.get("_src_hash") == tree, "profiles/pmc_busy.json was measured on another build: re-run tools/profile_round.sh"


def test_hdf5_container_without_h5py(tmp_path):
    """the "distances" dataset of --large h5py (bin/phyloligo.py:471-478: (N, N) float32, one dataset) without
    h5py - libhdf5 creates and sizes the file, the matrix is written into the dataset's contiguous data range by plain file
    writes (the path po_pwrite_rows takes), libhdf5 reads it back, and h5dump (the library's own tool) sees what h5py would."""
    import shutil
    import subprocess
    from phyloligo_amd import hdf5
    if not hdf5.available():
        pytest.skip("no libhdf5 >= 1.10 on this system")
    path = tmp_path / "d.h5"
    n = 37
    base = hdf5.create_f32_dataset(str(path), "distances", n, n)
    assert base > 0 and os.path.getsize(path) >= base + n * n * 4
    m = (np.arange(n * n, dtype=np.float32) / np.float32(7)).reshape(n, n)
    m[3, 5] = np.nan
    lib = _lib.load()
    fd = os.open(path, os.O_RDWR)
    try:                                                     # rows 0..19 whole, then rows 20.. as two column blocks
        _lib.check(lib.po_pwrite_rows(fd, m.ctypes.data, 20, n * 4, n * 4, base, n * 4, 3))
        left, right = np.ascontiguousarray(m[20:, :10]), np.ascontiguousarray(m[20:, 10:])
        _lib.check(lib.po_pwrite_rows(fd, left.ctypes.data, n - 20, 40, 40, base + 20 * n * 4, n * 4, 2))
        _lib.check(lib.po_pwrite_rows(fd, right.ctypes.data, n - 20, (n - 10) * 4, (n - 10) * 4, base + (20 * n + 10) * 4, n * 4, 2))
    finally:
        os.close(fd)
    back = hdf5.read_f32_dataset(str(path), "distances")
    assert back.shape == (n, n) and np.array_equal(back, m, equal_nan=True)
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        head = subprocess.run([h5dump, "-H", str(path)], capture_output=True, text=True, timeout=60).stdout
        assert 'DATASET "distances"' in head and "H5T_IEEE_F32LE" in head and "( %d, %d )" % (n, n) in head
    # ... and h5py itself, where some interpreter of this system has it (the image's /opt/conda python3.9 does: 3.3.0), reading
    # the file the way the reference's readers do (phyloselect.py:615-619: hf.get("distances"), then all of it)
    for exe in ("/opt/conda/bin/python3.9", "/opt/conda/bin/python"):
        if os.path.exists(exe) and subprocess.run([exe, "-c", "import h5py"], capture_output=True).returncode == 0:
            code = ("import h5py, numpy, sys\n"
                    "with h5py.File(sys.argv[1], 'r') as hf:\n"
                    "    d = hf.get('distances'); a = d[...]\n"
                    "    assert d.dtype == numpy.float32 and d.shape == (%d, %d) and d.chunks is None, (d.dtype, d.shape, d.chunks)\n"
                    "numpy.save(sys.argv[2], a)\n") % (n, n)
            r = subprocess.run([exe, "-c", code, str(path), str(tmp_path / "via_h5py.npy")], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr[-2000:]
            assert np.array_equal(np.load(tmp_path / "via_h5py.npy"), m, equal_nan=True)
            break
    with pytest.raises(OSError):
        hdf5.read_f32_dataset(str(path), "frequencies")
    assert hdf5.create_f32_dataset(str(tmp_path / "e.h5"), "distances", 0, 0) == 0
    assert hdf5.read_f32_dataset(str(tmp_path / "e.h5"), "distances").shape == (0, 0)

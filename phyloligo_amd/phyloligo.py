#!/usr/bin/env python3
"""Host-side mirror of the reference's dispatcher interface for the all-by-all distance path.

Same names, argument order and error behaviour as /root/reference/phylopackage/bin/phyloligo.py:
    compute_frequencies(...)  :980-997      compute_distances(...)  :536-553
    get_cmd()                 :1000-1034    main()                  :1036-1068
so a script written against the reference runs unchanged; every `--method` value executes the
HIP kernels of libphyloligo_amd.so (there is no joblib/scoop task farm and no CPU fallback).
"""
import argparse
import os
import sys
import time

import numpy as np

from . import api
from ._lib import METRICS, STRANDS

_ctx = None
LAST_STAGE2 = None       # PoStats of the first stage-2 call of the last compute_distances() (what --json-stats reports)
LAST_INGEST = None       # phase -> ms of the last compute_frequencies() when INGEST_PHASES is set (--json-stats, bench.py)
INGEST_PHASES = False    # time every step of compute_frequencies() (synchronises the device between steps)


def _timing(text):
    """PO_CLI_TIMING=1: phase times on stderr (tools/container_writer_bench.py compares the multi-rank writer with the
    single-process one); silent otherwise - the reference prints nothing of the kind."""
    if os.environ.get("PO_CLI_TIMING") == "1":
        sys.stderr.write("phyloligo_amd timing: %s\n" % text)


def _context():
    global _ctx
    if _ctx is None:
        _ctx = api.Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _ctx


class ProfileMatrix(np.ndarray):
    """float64[N, 4^k] frequencies exactly as compute_frequencies_joblib returns them
    (phyloligo.py:847-877), carrying the exact integer profile they were derived from."""

    def __new__(cls, freq, counts=None, totals=None, titles=None):
        obj = np.asarray(freq, dtype=np.float64).view(cls)
        obj.counts, obj.totals, obj.titles = counts, totals, titles
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.counts = self.totals = self.titles = None   # slices/copies are plain frequencies


def read_fasta(genome):
    """(seq uint8, offsets uint64[n+1], titles) of a multi-FASTA file, record order preserved."""
    size = os.path.getsize(genome)
    if size == 0:
        return np.zeros(0, np.uint8), np.zeros(1, np.uint64), []
    data = np.memmap(genome, dtype=np.uint8, mode="r")
    try:
        return api.fasta_index(data)
    finally:
        del data


def read_fasta_device(genome, phases=None):
    """The same records through the on-device parser: (seq CUDA uint8, offsets CUDA int64[n+1], titles), or None when
    the file has the one construct that parser leaves to the host (tabs on sequence lines) or is empty."""
    from ._lib import PhyloligoError, PO_EUNSUPPORTED
    if os.path.getsize(genome) == 0 or "torch" not in sys.modules:      # the device buffers of this path are torch tensors
        return None
    try:
        return api.fasta_index_dev(_context(), genome, phases)
    except PhyloligoError as exc:
        if exc.status == PO_EUNSUPPORTED:
            return None
        raise


def compute_frequencies(mthdrun, large, genome, pattern, strand, distchunksize=250, threads_max=4, workdir="."):
    """phyloligo.py:980-997.  Returns (frequencies, freq_name); freq_name is always None here
    (no on-disk frequency container: the count matrix lives in HBM / host memory)."""
    global LAST_INGEST
    if mthdrun not in ("joblib", "scoop", "hip"):
        print("Method {} is unknown".format(mthdrun), file=sys.stderr)      # :995, no exit
        return None, None
    if strand not in STRANDS:                                                # select_strand :146-148
        print("Error, strand parameter of selectd_strand() should be choose from {'both', 'minus', 'plus'}",
              file=sys.stderr)
        sys.exit(1)
    phases = {} if INGEST_PHASES else None
    t_last = [time.perf_counter()]

    def mark(name, sync=None):
        if phases is not None:
            if sync is not None:
                sync()
            now = time.perf_counter()
            phases[name] = phases.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now

    ctx = _context()
    mark("context_ms")
    ingest = read_fasta_device(genome, phases)
    mark("fasta_index_other_ms")
    if phases is not None:                 # (the steps inside were timed there: only what is left of the call counts here)
        phases["fasta_index_other_ms"] -= sum(phases.get(k, 0.0) for k in ("file_read_ms", "file_h2d_ms", "fasta_scan_ms", "device_alloc_ms",
                                                                             "fasta_extract_ms", "title_spans_d2h_ms"))
    if ingest is not None:
        # file bytes -> HBM -> records -> profiles -> frequencies without the sequence ever being walked on the host
        import torch
        d_seq, d_off, titles = ingest
        d_counts, d_totals = ctx.count_profiles(d_seq, d_off, pattern, strand)
        mark("stage1_ms", torch.cuda.synchronize)
        d_freq = ctx.frequencies(d_counts, d_totals)
        mark("count2freq_ms", torch.cuda.synchronize)
        freq = d_freq.cpu().numpy()
        mark("frequencies_d2h_ms")
        counts = d_counts.cpu().numpy().view(np.uint32)
        totals = d_totals.cpu().numpy().view(np.uint64)
        mark("counts_d2h_ms")
    else:
        seq, offsets, titles = read_fasta(genome)
        mark("host_parse_ms")
        counts, totals = ctx.count_profiles(seq, offsets, pattern, strand)
        mark("stage1_host_pointers_ms")
        freq = ctx.frequencies(counts, totals)
        mark("count2freq_host_pointers_ms")
    result = ProfileMatrix(freq, counts, totals, titles)
    mark("wrap_ms")
    LAST_INGEST = phases
    return result, None


def _reserve_file(fd, size):
    """Size the container and, where the filesystem can, allocate its blocks up front: parallel pwrite into a preallocated
    file measured 12.2-12.6 GB/s against 10.5-10.9 GB/s into a merely truncated one (gpurun box, overlay on ext4, 4 GB,
    tools/exp/pwrite_rate.py).  fallocate(2) itself, not posix_fallocate: glibc's emulation for filesystems without
    it writes into every block."""
    if size > 0:
        try:
            import ctypes
            libc = ctypes.CDLL(None, use_errno=True)
            libc.fallocate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
            libc.fallocate.restype = ctypes.c_int
            if libc.fallocate(fd, 0, 0, size) == 0:
                return
        except (OSError, AttributeError):
            pass
    if os.fstat(fd).st_size < size:          # never shorten: an HDF5 file keeps whatever the library has put behind the data
        os.ftruncate(fd, size)


def _open_raw_container(out_file, size):
    """A raw container of exactly `size` bytes, every one of which the caller is going to write.  An existing file is reused, not
    truncated to nothing first: O_TRUNC on a 10 GB file whose pages sit in the page cache took 0.72 - 0.78 s on the gpurun box
    (as long as removing it), overwriting it in place nothing beyond the write (tools/exp/overwrite_cost.py) - a second run of the
    same command went from 2.1 s to 1.4 s.  Only a longer file is cut back, to the new size."""
    fd = os.open(out_file, os.O_RDWR | os.O_CREAT, 0o666)
    try:
        if os.fstat(fd).st_size > size:
            os.ftruncate(fd, size)
        _reserve_file(fd, size)
    except BaseException:
        os.close(fd)
        raise
    return fd


def _pwrite_rows(fd, host, row0, col0, n, threads=8, base=0):
    """host[R, C] float32 (unit inner stride) -> rows row0.., columns col0.. of the n x n float32 matrix that starts at byte
    `base` of the file behind fd (0: the raw container; the data offset of the "distances" dataset: the HDF5 container)
    (po_pwrite_rows: parallel pwrite in C, no Python per row)."""
    import ctypes
    from . import _lib
    rows, cols = host.shape
    if rows == 0 or cols == 0:
        return
    assert host.dtype == np.float32 and host.strides[1] == 4 and host.strides[0] >= cols * 4
    _lib.check(_lib.load().po_pwrite_rows(fd, ctypes.c_void_p(host.ctypes.data), rows, cols * 4, host.strides[0],
                                          base + (row0 * n + col0) * 4, n * 4, threads))


def _bc_memmap_diagonal(buf, lo, hi, empty):
    """The reference's memmap variant evaluates Bray-Curtis through BC_loc -> phylodist.BC -> pairwise_distances(X, X[s],
    metric="braycurtis") (bin/phyloligo.py:214-217, core/phylodist.py:76-79), i.e. SciPy cdist: an EMPTY profile against
    itself is 0/0 = nan there, where the joblib variant (pdist + squareform, :381) has an exact 0 on the diagonal
    (tests/golden/memmap.npz vs distances.npz).  The container follows the variant it stands for."""
    for i in empty:
        if lo <= i < hi:
            buf[i - lo, i] = np.nan


def _write_raw_f32(out_file, n, rows, writers=8, fix=None, base=None):
    """The container of compute_distances_memmap (phyloligo.py:394-427): headerless row-major float32[n, n] (:413), the
    file phyloligo_comparemat.py:16-24 and phyloselect.py:606-614 read back.  Row blocks come off the device into
    one of two reusable host buffers and go to the file through po_pwrite_rows (parallel pwrite) on a background thread
    while the next block is computed and copied: no second ndarray, no page faults on a file mapping (numpy.memmap
    assignment measured ~3 GB/s).  fix(buf, lo, hi): host-side touch-up of a finished row block before it is written."""
    import concurrent.futures as cf
    # base is None: a new raw container.  base = byte offset: the matrix goes into an EXISTING file from there on (the HDF5
    # container: created and sized by libhdf5, phyloligo_amd/hdf5.py)
    fd = os.open(out_file, os.O_RDWR) if base is not None else _open_raw_container(out_file, n * n * 4)
    try:
        # blocks allocated up front either way (fallocate keeps what libhdf5 has already written at the head of its file)
        if base is not None:
            _reserve_file(fd, base + n * n * 4)
        if n == 0:
            return
        step = min(n, _row_chunk(n, 4, budget=512 << 20))
        bufs = [np.empty((step, n), dtype=np.float32) for _ in range(2 if step < n else 1)]
        pending = [None for _ in bufs]
        with cf.ThreadPoolExecutor(max_workers=1) as pool:       # one submitter; the parallelism is inside po_pwrite_rows
            for k, lo in enumerate(range(0, n, step)):
                hi = min(n, lo + step)
                which = k % len(bufs)
                if pending[which] is not None:                   # the block written from this buffer two rounds ago
                    pending[which].result()
                buf = bufs[which][:hi - lo]
                rows(lo, hi, "float32", lo == 0 and hi == n, out=buf)
                if fix is not None:
                    fix(buf, lo, hi)
                pending[which] = pool.submit(_pwrite_rows, fd, buf, lo, 0, n, writers, base or 0)
            for f in pending:
                if f is not None:
                    f.result()
    finally:
        os.close(fd)


class _BlockWriter:
    """Device blocks of one rank -> their byte ranges of the shared float32 container (the multi-GPU form of
    _write_raw_f32): two pinned host buffers; the device-to-host copy of chunk k + 1 runs on a side stream while
    po_pwrite_rows (parallel pwrite) puts chunk k into the file from a background thread."""

    def __init__(self, fd, n, device, chunk_bytes=256 << 20, writers=8, base=0):
        import concurrent.futures as cf
        import torch
        self.torch, self.fd, self.n, self.writers, self.base = torch, fd, n, writers, base
        self.chunk = chunk_bytes
        self.stage = [torch.empty(chunk_bytes // 4, dtype=torch.float32, pin_memory=True) for _ in range(2)]
        self.pending = [None, None]
        self.stream = torch.cuda.Stream(device)
        self.pool = cf.ThreadPoolExecutor(max_workers=1)
        self.k = 0

    def put(self, t, row0, col0, fix=None):
        """t[R, C] float32 on the device -> rows row0.., columns col0.. of the file"""
        torch = self.torch
        R, C = t.shape
        if R == 0 or C == 0:
            return
        per = max(1, self.chunk // (C * 4))
        self.stream.wait_stream(torch.cuda.current_stream(t.device))
        for a in range(0, R, per):
            b = min(R, a + per)
            which = self.k & 1
            self.k += 1
            if self.pending[which] is not None:                  # the chunk written from this buffer two rounds ago
                self.pending[which].result()
            view = self.stage[which][:(b - a) * C].view(b - a, C)
            with torch.cuda.stream(self.stream):
                view.copy_(t[a:b], non_blocking=True)
            self.stream.synchronize()                            # (the previous chunk is being written meanwhile)
            host = view.numpy()
            if fix is not None:
                fix(host, row0 + a, row0 + b)
            self.pending[which] = self.pool.submit(_pwrite_rows, self.fd, host, row0 + a, col0, self.n, self.writers, self.base)

    def close(self):
        try:
            for f in self.pending:
                if f is not None:
                    f.result()
        finally:
            self.pool.shutdown(wait=True)


_SINGLE_CALL_BYTES = 4 << 30      # larger float64 matrices are computed and copied in row blocks


def _row_chunk(n, itemsize, budget=1 << 30):
    return max(128, (budget // max(1, n * itemsize)) // 128 * 128)


def compute_distances(mthdrun, large, frequencies, freq_name, out_file, dist, threads_max=4, freqchunksize=250,
                      workdir="."):
    """phyloligo.py:536-553.  `--large None`: returns the float64[N,N] matrix (main writes it).
    `--large memmap`: writes the headerless float32 row-major matrix to out_file as
    compute_distances_memmap does (:394-427, container dtype :413) and returns None.
    `--large h5py`: writes the HDF5 file of compute_distances_h5py / join_distance_results (:456-534): one float32 dataset
    "distances" of shape (N, N), and returns None."""
    if mthdrun not in ("joblib", "scoop", "hip"):
        print("Error, method {} is not implemented for pairwise distances computation".format(mthdrun), file=sys.stderr)
        return None
    if dist not in METRICS:                                                  # :383-385
        print("Error, unknown metric methodfor joblib: {}".format(dist), file=sys.stderr)
        sys.exit(1)
    if large == "h5py" and mthdrun != "scoop":
        from . import hdf5
        if not hdf5.available():
            print("Error, --large h5py needs libhdf5 (>= 1.10), which was not found on this system", file=sys.stderr)
            sys.exit(1)
    ctx = _context()
    counts = getattr(frequencies, "counts", None)
    totals = getattr(frequencies, "totals", None)
    n = frequencies.shape[0]
    if mthdrun == "scoop":                   # compute_distances_scoop (:313-362) has no --large variants
        large = "None"
    # The reference computes from the array it is handed.  The integer profiles riding on a ProfileMatrix are only
    # a shortcut while they still ARE that array: in-place edits (frequencies *= w, frequencies[mask] = 0, ...)
    # keep the attributes, so check count/total == array bit for bit and otherwise go by the array's values.
    if counts is not None and (counts.shape != frequencies.shape or
                               not np.array_equal(ctx.frequencies(counts, totals), np.asarray(frequencies))):
        counts = totals = None

    global LAST_STAGE2
    LAST_STAGE2 = None

    def rows(lo, hi, dtype, symmetric, out=None):
        global LAST_STAGE2
        if counts is not None:
            res, st = ctx.pairwise(counts, totals, dist, lo, hi, dtype=dtype, symmetric=symmetric, out=out, want_stats=True)
        else:
            res, st = ctx.pairwise_freq(np.asarray(frequencies, dtype=np.float64), dist, lo, hi, dtype=dtype,
                                        symmetric=symmetric, out=out, want_stats=True)
        if LAST_STAGE2 is None:
            LAST_STAGE2 = dict(st, rows=[int(lo), int(hi)])
        return res

    if large in ("memmap", "h5py"):
        fix = None
        if dist == "BC":                     # BC_loc and BC_h5py both go through phylodist.BC (SciPy cdist): see _bc_memmap_diagonal
            empty = np.flatnonzero(~np.asarray(frequencies).any(axis=1))
            if empty.size:
                fix = lambda buf, lo, hi: _bc_memmap_diagonal(buf, lo, hi, empty)      # noqa: E731
        base = None
        if large == "h5py":
            # join_distance_results (:456-478): ONE dataset "distances", (N, N), float32.  libhdf5 creates and sizes the file; the
            # matrix then goes into the dataset's contiguous data range exactly as it goes into the raw container
            base = hdf5.create_f32_dataset(out_file, "distances", n, n)
        _write_raw_f32(out_file, n, rows, fix=fix, base=base)
        return None
    if n * n * 8 <= _SINGLE_CALL_BYTES:
        return rows(0, n, "float64", True)
    res = np.empty((n, n), dtype=np.float64)
    step = _row_chunk(n, 8)
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        rows(lo, hi, "float64", False, out=res[lo:hi])          # straight into the result: no per-block ndarray
    return res


def get_cmd(argv=None):
    """phyloligo.py:1000-1034, option for option.  -k and -p share dest="pattern": whichever comes
    last on the command line wins; with neither the int default 4 is used (-> "1111")."""
    parser = argparse.ArgumentParser(prog="phyloligo.py")
    parser.add_argument("-i", "--assembly", action="store", required=True, dest="genome",
                        help="multifasta of the genome assembly")
    parser.add_argument("-k", "--lgMot", action="store", dest="pattern", default=4, type=int,
                        help="word lenght / kmer length / k [default:%(default)d]")
    parser.add_argument("-s", "--strand", action="store", dest="strand", default="both",
                        choices=["both", "plus", "minus"],
                        help="strand used to compute microcomposition. [default:%(default)s]")
    parser.add_argument("-d", "--distance", action="store", dest="dist", default="Eucl",
                        choices=["Eucl", "JSD", "KT", "BC", "SC"],
                        help="how to compute distance between two signatures : Eucl : Euclidean[default:%(default)s], "
                             "JSD : Jensen-Shannon divergence, KT: Kendall's tau, BC: Bray-Curtis, SC:Spearman Correlation")
    parser.add_argument("--freq-chunk-size", action="store", dest="freqchunksize", type=int, default=250,
                        help="accepted for compatibility (scoop chunking; unused on the GPU)")
    parser.add_argument("--dist-chunk-size", action="store", dest="distchunksize", type=int, default=250,
                        help="accepted for compatibility (scoop chunking; unused on the GPU)")
    parser.add_argument("--method", action="store", choices=["scoop", "joblib", "hip"], default="joblib",
                        dest="mthdrun", required=True,
                        help="kept from the reference; every value runs the MI355X HIP path")
    parser.add_argument("--large", action="store", dest="large", choices=["None", "memmap", "h5py"], default="None",
                        help="memmap: write the matrix as raw float32 instead of text")
    parser.add_argument("-c", "--cpu", action="store", dest="threads_max", type=int, default=4,
                        help="accepted for compatibility (host threads are not the compute resource)")
    parser.add_argument("-o", "--out", action="store", dest="out_file", default="phyloligo.out",
                        help="output file[default:%(default)s]")
    parser.add_argument("-q", "--outfreq", action="store", dest="out_freq_file",
                        help="kmer frequencies output file")
    parser.add_argument("-w", "--workdir", action="store", dest="workdir", default=".", help="working directory")
    parser.add_argument("-p", "--pattern", action="store", dest="pattern", default="1111",
                        help="spaced-word pattern string, only containing 1s and 0s, i.e. '100101001', default='1111'")
    parser.add_argument("--json-stats", action="store", dest="json_stats", default=None,
                        help="not in the reference (SURVEY section 5): write sizes, phase times and the stage-2 kernel of this run "
                             "to this file as JSON; the five progress lines on stdout stay as they are")
    parser.add_argument("--gpus", action="store", dest="gpus", type=int, default=1,
                        help="not in the reference: GPUs of this node to use, one process each (python -m phyloligo_amd starts "
                             "the ranks itself; the analogue of the reference fanning out to -c joblib workers) [default:%(default)d]")
    params = parser.parse_args(argv)
    params.workdir = os.path.abspath(params.workdir)
    return params


def main_distributed(params):
    """The same job on the GPUs of one node, one process per GPU (launched by torch.distributed.run: WORLD_SIZE > 1).

    The reference spreads row slices over joblib workers that all see the frequency matrix (gen_even_slices,
    phyloligo.py:424,516).  Here every rank profiles its own block of contigs, ONE all-gather moves the exact count
    matrix, the block grid is dealt out like a round-robin tournament (dist.RowBlockPlan: every unordered pair evaluated
    once in the whole job), the transposed blocks go to the ranks whose rows they complete, and every rank writes its
    rows of the output: straight into its byte range of the float32 container (--large memmap), or in rank order into the
    text matrix.  PO_CLI_REHEARSAL=1: all ranks on GPU 0 with gloo collectives (a one-GPU box can run the code path)."""
    import torch
    import torch.distributed as tdist
    from .dist import RowBlockPlan
    t_start = time.perf_counter()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rehearsal = os.environ.get("PO_CLI_REHEARSAL") == "1"
    local = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
    if params.mthdrun not in ("joblib", "hip"):
        if rank == 0:
            print("Error, the multi-GPU run supports --method joblib|hip", file=sys.stderr)
        sys.exit(1)
    if params.large == "h5py":
        from . import hdf5
        if not hdf5.available():
            if rank == 0:
                print("Error, --large h5py needs libhdf5 (>= 1.10), which was not found on this system", file=sys.stderr)
            sys.exit(1)
    if params.strand not in STRANDS or params.dist not in METRICS:
        sys.exit(1)
    from .dist import first_contact
    first_contact("cuda.set_device", torch.cuda.set_device, local)
    if rehearsal:
        first_contact("init_process_group", tdist.init_process_group, "gloo")
    else:
        first_contact("init_process_group", tdist.init_process_group, "nccl", device_id=torch.device("cuda", local))
    global _ctx
    _ctx = api.Context(local)
    ctx, dev = _ctx, torch.device("cuda", local)
    say = print if rank == 0 else (lambda *a, **k: None)
    say("Using pattern {}".format(params.pattern))
    if rank == 0 and not os.path.isdir(params.workdir):
        os.makedirs(params.workdir)
    say("Computing frequencies")
    ingest = read_fasta_device(params.genome)
    if ingest is not None:
        d_seq, d_off, _ = ingest
    else:
        seq, offsets, _ = read_fasta(params.genome)
        d_seq, d_off = torch.from_numpy(seq).to(dev), torch.from_numpy(offsets.astype(np.int64)).to(dev)
    n = int(d_off.numel()) - 1
    plan = RowBlockPlan(n, world)
    lo, hi = plan.rows(rank)
    my_off = (d_off[lo:hi + 1] - d_off[lo]).contiguous()
    my_seq = d_seq[int(d_off[lo]):int(d_off[hi])]
    if my_seq.data_ptr() % 16:                                  # the kernels read aligned 16-byte vectors
        my_seq = my_seq.clone()
    my_counts, my_totals = ctx.count_profiles(my_seq, my_off, params.pattern, params.strand)
    if rehearsal:
        counts, totals = first_contact("all_gather_profiles", plan.all_gather_profiles, my_counts.cpu(), my_totals.cpu(), tdist)
        counts, totals = counts.to(dev), totals.to(dev)
    else:
        def gather():                     # the ranks' first collective: a failure surfaces at the synchronize at the latest
            c, t = plan.all_gather_profiles(my_counts, my_totals, tdist)
            torch.cuda.synchronize(dev)
            return c, t
        counts, totals = first_contact("all_gather_profiles", gather)
    say("Computing Pairwise distances")
    t_dist0 = time.perf_counter()
    dtype = torch.float32 if params.large in ("memmap", "h5py") else torch.float64
    slab, mirrors = plan.allocate(rank, dev, dtype)
    plan.compute(ctx, counts, totals, params.dist, rank, slab, mirrors)
    torch.cuda.synchronize(dev)
    t_computed = time.perf_counter()
    if params.large in ("memmap", "h5py"):
        # The row-completing exchange runs first (point to point over xGMI: cheap next to any file system), so that every
        # rank holds ITS ROWS of the matrix whole and writes one contiguous byte range of the file in large pieces.  Measured
        # on the gpurun box (tools/ubench/container_write_modes.cpp, page cache of overlay/ext4, 3.6 GB): buffered writes
        # take the inode lock, so ranks that each pwrite their 60 kB pieces of every row reach 4.7 GB/s between two of them and
        # 2.5 GB/s between eight (the lock changes hands at every call; round 3 did exactly that, 0.82 s for this file),
        # a shared mapping written with memcpy 0.6 - 2.7 GB/s, while contiguous row slabs in large pieces reach 11 GB/s from
        # two and 15.6 GB/s from eight processes.
        if params.out_freq_file and rank == 0:
            print("Writing frequency matrix")
            api.write_mat_text(params.out_freq_file, ctx.frequencies(counts, totals).cpu().numpy())
        base = [0]
        if rank == 0:
            if params.large == "h5py":                          # libhdf5 creates and sizes the file; the ranks fill the dataset's data range
                base[0] = hdf5.create_f32_dataset(params.out_file, "distances", n, n)
            else:
                os.close(_open_raw_container(params.out_file, n * n * 4))
        tdist.broadcast_object_list(base, src=0)                # where the matrix starts in the file
        empty = (totals.cpu().numpy() == 0) if params.dist == "BC" else None

        def diag_fix(host, a, b):                               # rows [a, b) of the matrix, whole
            _bc_memmap_diagonal(host, a, b, np.flatnonzero(empty[a:b]) + a)

        fix = diag_fix if (empty is not None and empty[lo:hi].any()) else None
        if rehearsal:                                           # gloo moves host tensors; the slab stays on the device
            mirrors_h = [None if m is None else m.cpu() for m in mirrors]
            del mirrors
            plan.complete_rows(rank, slab, mirrors_h, tdist, stage_device="cpu")
            del mirrors_h
        else:
            plan.complete_rows(rank, slab, mirrors, tdist)
            del mirrors
        torch.cuda.synchronize(dev)
        t_exchanged = time.perf_counter()
        tdist.barrier()                                         # the file exists and has its size
        fd = os.open(params.out_file, os.O_RDWR)
        # (two pinned chunks of at most 256 MB; pinning costs ~0.07 s per GB, a small job does not pay for more than its slab)
        writer = _BlockWriter(fd, n, dev, chunk_bytes=min(256 << 20, max(1 << 20, (hi - lo) * n * 4)), base=int(base[0]))
        try:
            writer.put(slab, lo, 0, fix=fix)                    # device -> pinned host chunks -> parallel pwrite, overlapped
        finally:
            try:
                writer.close()
            finally:
                os.close(fd)
        tdist.barrier()
        t_end = time.perf_counter()
        _timing("rank %d of %d: distances + container %.3f s (allocate + compute %.3f, row-completing exchange %.3f, copy + write %.3f)"
                % (rank, world, t_end - t_dist0, t_computed - t_dist0, t_exchanged - t_computed, t_end - t_exchanged))
        if params.json_stats and rank == 0:
            _write_json_stats(params, _Shape(n, int(counts.shape[1])), t_dist0 - t_start, t_end - t_dist0, 0.0, t_end - t_start, gpus=world)
        tdist.destroy_process_group()
        return 0
    # text matrix: rows have to be complete - the transposed blocks go to the ranks whose rows they belong to
    if rehearsal:                                               # gloo moves host tensors
        slab_h, mirrors_h = slab.cpu(), [None if m is None else m.cpu() for m in mirrors]
        del slab, mirrors
        slab = plan.complete_rows(rank, slab_h, mirrors_h, tdist)
    else:
        plan.complete_rows(rank, slab, mirrors, tdist)
        torch.cuda.synchronize(dev)
        del mirrors
    step = max(1, (256 << 20) // max(1, n * slab.element_size()))     # rows per host copy: the slab never sits on the host whole

    def row_chunks():
        for a in range(0, hi - lo, step):
            yield a, slab[a:min(hi - lo, a + step)].cpu().numpy()

    if params.out_freq_file and rank == 0:
        print("Writing frequency matrix")
        api.write_mat_text(params.out_freq_file, ctx.frequencies(counts, totals).cpu().numpy())
    say("Writing distance matrix")
    for r in range(world):                                      # rank order = row order
        if r == rank:
            if r == 0 and hi == lo:
                api.write_mat_text(params.out_file, np.zeros((0, n)))
            for a, rows in row_chunks():
                api.write_mat_text(params.out_file, rows, append=(r > 0 or a > 0))
        tdist.barrier()
    if params.json_stats and rank == 0:
        t_end = time.perf_counter()
        _write_json_stats(params, _Shape(n, int(counts.shape[1])), t_dist0 - t_start, t_computed - t_dist0, t_end - t_computed,
                          t_end - t_start, gpus=world)
    tdist.destroy_process_group()
    return 0


def main(argv=None):
    params = get_cmd(argv)
    if type(params.pattern) == int:                      # :1040-1041
        params.pattern = str("1") * params.pattern
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return main_distributed(params)
    t_start = time.perf_counter()
    print("Using pattern {}".format(params.pattern))
    if not os.path.isdir(params.workdir):
        os.makedirs(params.workdir)
    print("Computing frequencies")
    global INGEST_PHASES
    INGEST_PHASES = bool(params.json_stats)              # the phase times of the ingest go into the JSON record
    frequencies, freq_name = compute_frequencies(params.mthdrun, params.large, params.genome, params.pattern,
                                                 params.strand, params.distchunksize, params.threads_max,
                                                 params.workdir)
    t_freq = time.perf_counter()
    print("Computing Pairwise distances")
    t_dist0 = time.perf_counter()
    res = compute_distances(params.mthdrun, params.large, frequencies, freq_name, params.out_file, params.dist,
                            params.threads_max, params.freqchunksize, params.workdir)
    t_dist = time.perf_counter()
    _timing("single process: distances%s %.3f s" % (" + container" if params.large in ("memmap", "h5py") else "", t_dist - t_dist0))
    if params.out_freq_file:
        print("Writing frequency matrix")
        api.write_mat_text(params.out_freq_file, np.asarray(frequencies))
    if not (params.mthdrun in ("joblib", "hip") and params.large != "None"):      # :1064
        print("Writing distance matrix")
        if res is not None:                  # None: unknown --method, nothing was computed (:552 only prints)
            api.write_mat_text(params.out_file, res)
    if params.json_stats and frequencies is not None:
        _write_json_stats(params, frequencies, t_freq - t_start, t_dist - t_dist0, time.perf_counter() - t_dist, time.perf_counter() - t_start)
    return 0


class _Shape:
    """stands in for the frequency matrix where only its shape is wanted (--json-stats of a multi-rank run)"""

    def __init__(self, n, dim):
        self.shape = (n, dim)


def _write_json_stats(params, frequencies, freq_s, dist_s, write_s, total_s, gpus=1):
    """--json-stats ("keep the same five lines for CLI fidelity; add --json-stats")"""
    import json
    from . import _lib
    n, dim = (int(x) for x in frequencies.shape)
    stats = {"library": _lib.load().po_version().decode(), "device": _context().device_name, "gpus": gpus,
             "assembly": os.path.abspath(params.genome), "contigs": n, "words": dim, "pattern": str(params.pattern),
             "strand": params.strand, "metric": params.dist, "large": params.large, "pairs": n * (n - 1) // 2,
             "seconds": {"frequencies": freq_s, "distances" + ("_and_container" if params.large in ("memmap", "h5py") else ""): dist_s,
                         "writing": write_s, "total": total_s},
             "stage2_first_call": LAST_STAGE2,
             "ingest_phases_ms": None if LAST_INGEST is None else {k: round(v, 3) for k, v in LAST_INGEST.items()}}
    with open(params.json_stats, "w") as fh:
        json.dump(stats, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
    sys.exit(0)

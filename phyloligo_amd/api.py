"""Thin Python face of the C ABI: numpy arrays (host entry points) or torch CUDA tensors
(device entry points) in, the same out.  All arithmetic happens in libphyloligo_amd.so.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (METRICS, STRANDS, PO_F32, PO_F64, PO_FLAG_NO_SYMMETRY, PO_FLAG_NO_TABLE_PATH, PO_FLAG_NO_RC_FOLD, PO_FLAG_PAIRDOT_I8, PO_FLAG_NO_PAIRDOT, PoBlock, PoStats,
                   check)


def normalise_pattern(pattern):
    """-k N and -p PATTERN share one destination in the reference CLI
    (bin/phyloligo.py:1006,1027); an int means the contiguous pattern "1"*N (:1040-1041)."""
    if isinstance(pattern, (int, np.integer)) and not isinstance(pattern, bool):
        pattern = "1" * int(pattern)
    return str(pattern)


def pattern_info(pattern):
    lib = _lib.load()
    w, k, d = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
    check(lib.po_pattern_info(normalise_pattern(pattern).encode(), ctypes.byref(w), ctypes.byref(k), ctypes.byref(d)))
    return w.value, k.value, d.value


def device_count():
    return _lib.load().po_device_count()


def _np_ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Context:
    """One po_ctx = one GPU.  Device-tensor calls run on torch's current stream."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self.device = int(device)
        check(self._lib.po_ctx_create(ctypes.byref(self._h), self.device))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.po_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        check(self._lib.po_ctx_device_name(self._h, buf, 256))
        return buf.value.decode()

    def synchronize(self):
        check(self._lib.po_ctx_synchronize(self._h))

    def trim(self):
        """free the device workspaces this context has grown (po_ctx_trim); the next call allocates again"""
        check(self._lib.po_ctx_trim(self._h))

    def _use_torch_stream(self):
        import torch
        check(self._lib.po_ctx_set_stream(self._h, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    # ---- stage 1 ---------------------------------------------------------------------------
    def count_profiles(self, seq, offsets, pattern="1111", strand="both"):
        """seq: uint8 concatenated sequence bytes, offsets: uint64[n+1].  numpy in -> numpy out
        (uint32 counts[n, 4^k], uint64 totals[n]); torch CUDA tensors in -> torch CUDA tensors out."""
        pat = normalise_pattern(pattern)
        _, _, dim = pattern_info(pat)
        if strand not in STRANDS:
            raise _lib.PhyloligoError(_lib.PO_EINVAL, "strand must be one of both/plus/minus (got %r)" % (strand,))
        if _is_torch(seq):
            import torch
            n = offsets.numel() - 1
            assert seq.dtype == torch.uint8 and seq.is_cuda and seq.is_contiguous()
            assert offsets.dtype == torch.int64 and offsets.is_cuda and offsets.is_contiguous()
            if seq.device.index != self.device or offsets.device != seq.device:
                raise _lib.PhyloligoError(_lib.PO_EINVAL, "tensors are on %s / %s, this context drives cuda:%d"
                                          % (seq.device, offsets.device, self.device))
            counts = torch.empty((n, dim), dtype=torch.int32, device=seq.device)
            totals = torch.empty((n,), dtype=torch.int64, device=seq.device)
            self._use_torch_stream()
            check(self._lib.po_count_profiles_dev(self._h, seq.data_ptr(), offsets.data_ptr(), n, seq.numel(),
                                                  pat.encode(), STRANDS[strand], counts.data_ptr(), totals.data_ptr()))
            return counts, totals
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.shape[0] - 1
        counts = np.zeros((n, dim), dtype=np.uint32)
        totals = np.zeros((n,), dtype=np.uint64)
        check(self._lib.po_count_profiles(self._h, _np_ptr(seq), _np_ptr(offsets), n, pat.encode(), STRANDS[strand],
                                          _np_ptr(counts), _np_ptr(totals)))
        return counts, totals

    def count_profiles_ranges(self, seq, begins, ends, pattern="1111", strand="both"):
        """Profiles of arbitrary (overlapping) byte ranges [begins[i], ends[i]) of one sequence buffer --
        Kount.py's sliding windows.  numpy in -> numpy out; torch CUDA tensors in -> torch CUDA tensors out."""
        pat = normalise_pattern(pattern)
        _, _, dim = pattern_info(pat)
        if strand not in STRANDS:
            raise _lib.PhyloligoError(_lib.PO_EINVAL, "strand must be one of both/plus/minus (got %r)" % (strand,))
        if _is_torch(seq):
            import torch
            self._check_device(seq, begins, ends)
            assert seq.dtype == torch.uint8 and seq.is_contiguous()
            assert begins.dtype == torch.int64 and ends.dtype == torch.int64 and begins.is_contiguous() and ends.is_contiguous()
            n = begins.numel()
            counts = torch.empty((n, dim), dtype=torch.int32, device=seq.device)
            totals = torch.empty((n,), dtype=torch.int64, device=seq.device)
            sum_lengths = int((ends - begins).sum().item()) if n else 0
            self._use_torch_stream()
            check(self._lib.po_count_profiles_ranges_dev(self._h, seq.data_ptr(), seq.numel(), begins.data_ptr(), ends.data_ptr(), n,
                                                         sum_lengths, pat.encode(), STRANDS[strand], counts.data_ptr(),
                                                         totals.data_ptr()))
            return counts, totals
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        begins = np.ascontiguousarray(begins, dtype=np.uint64)
        ends = np.ascontiguousarray(ends, dtype=np.uint64)
        n = begins.shape[0]
        counts = np.zeros((n, dim), dtype=np.uint32)
        totals = np.zeros((n,), dtype=np.uint64)
        check(self._lib.po_count_profiles_ranges(self._h, _np_ptr(seq), seq.shape[0], _np_ptr(begins), _np_ptr(ends), n,
                                                 pat.encode(), STRANDS[strand], _np_ptr(counts), _np_ptr(totals)))
        return counts, totals

    def _check_device(self, *tensors):
        for t in tensors:
            if t.device.type != "cuda" or t.device.index != self.device:
                raise _lib.PhyloligoError(_lib.PO_EINVAL, "tensor is on %s, this context drives cuda:%d" % (t.device, self.device))

    def count_byte_ranges(self, seq, begins, ends, byte):
        """Occurrences of one byte value in every range (device tensors): the numerator of Kount.py's N gate."""
        import torch
        self._check_device(seq, begins, ends)
        assert seq.dtype == torch.uint8 and begins.dtype == torch.int64 and ends.dtype == torch.int64
        out = torch.empty((begins.numel(),), dtype=torch.int64, device=seq.device)
        self._use_torch_stream()
        check(self._lib.po_count_byte_ranges_dev(self._h, seq.data_ptr(), seq.numel(), begins.data_ptr(), ends.data_ptr(),
                                                 begins.numel(), int(byte), out.data_ptr()))
        return out

    def profile_distances(self, counts, totals, proto, metric="JSD"):
        """Distance of every profile to ONE prototype frequency vector (Kount.py's JSD / KL / Eucl, unscaled)."""
        code = {"Eucl": 0, "JSD": 1, "KL": _lib.PO_KL}.get(metric)
        if code is None:
            raise _lib.PhyloligoError(_lib.PO_EINVAL, "metric must be JSD, KL or Eucl (got %r)" % (metric,))
        if _is_torch(counts):
            import torch
            self._check_device(counts, totals)
            n, dim = counts.shape
            assert counts.dtype == torch.int32 and totals.dtype == torch.int64 and counts.is_contiguous()
            d_proto = proto if _is_torch(proto) else torch.from_numpy(np.ascontiguousarray(proto, dtype=np.float64)).to(counts.device)
            assert d_proto.dtype == torch.float64 and d_proto.numel() == dim
            out = torch.empty((n,), dtype=torch.float64, device=counts.device)
            self._use_torch_stream()
            check(self._lib.po_profile_distances_dev(self._h, counts.data_ptr(), totals.data_ptr(), n, dim, d_proto.data_ptr(),
                                                     code, out.data_ptr()))
            return out
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        totals = np.ascontiguousarray(totals, dtype=np.uint64)
        proto = np.ascontiguousarray(proto, dtype=np.float64)
        n, dim = counts.shape
        assert proto.shape == (dim,)
        out = np.zeros(n, dtype=np.float64)
        check(self._lib.po_profile_distances(self._h, _np_ptr(counts), _np_ptr(totals), n, dim, _np_ptr(proto), code,
                                             _np_ptr(out)))
        return out

    def frequencies(self, counts, totals):
        if _is_torch(counts):
            import torch
            n, dim = counts.shape
            out = torch.empty((n, dim), dtype=torch.float64, device=counts.device)
            self._use_torch_stream()
            check(self._lib.po_frequencies_dev(self._h, counts.data_ptr(), totals.data_ptr(), n, dim, out.data_ptr()))
            return out
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        totals = np.ascontiguousarray(totals, dtype=np.uint64)
        n, dim = counts.shape
        out = np.zeros((n, dim), dtype=np.float64)
        check(self._lib.po_frequencies(self._h, _np_ptr(counts), _np_ptr(totals), n, dim, _np_ptr(out)))
        return out

    # ---- stage 2 ---------------------------------------------------------------------------
    def reserve(self, n, dim, metric):
        check(self._lib.po_pairwise_reserve(self._h, n, dim, METRICS[metric]))

    def pairwise(self, counts, totals, metric="Eucl", row_begin=0, row_end=None, dtype="float64", symmetric=True,
                 out=None, want_stats=False, table_path=True, rc_fold=True, pairdot_i8=False, pairdot=True):
        """Rows [row_begin,row_end) x all columns of the distance matrix from integer profiles.
        table_path=False forces the general JSD kernel even for record blocks with equal totals;
        rc_fold=False keeps every word even when the profiles are reverse-complement symmetric;
        pairdot_i8=True keeps the materialised KT / BC operand as int8 instead of FP4 (same results);
        pairdot=False leaves KT / BC to the vector-ALU kernels (SAD / O(D^2) Kendall)."""
        return self._pairwise(counts, totals, None, metric, row_begin, row_end, dtype, symmetric, out, want_stats,
                              (0 if table_path else PO_FLAG_NO_TABLE_PATH) | (0 if rc_fold else PO_FLAG_NO_RC_FOLD) |
                              (PO_FLAG_PAIRDOT_I8 if pairdot_i8 else 0) | (0 if pairdot else PO_FLAG_NO_PAIRDOT))

    def pairwise_freq(self, freq, metric="Eucl", row_begin=0, row_end=None, dtype="float64", symmetric=True,
                      out=None, want_stats=False, rc_fold=True, table_path=True):
        """The same from a float64 frequency matrix (the reference's `frequencies` argument).  Frequencies that are
        count / total bit for bit (count2freq output) are traced back to the integer profiles on the device and take
        the same kernels as `pairwise`; table_path=False keeps the general float64 kernels."""
        return self._pairwise(None, None, freq, metric, row_begin, row_end, dtype, symmetric, out, want_stats,
                              (0 if rc_fold else PO_FLAG_NO_RC_FOLD) | (0 if table_path else PO_FLAG_NO_TABLE_PATH))

    def pairwise_blocks(self, counts, totals, metric, blocks, dtype="float64", want_stats=False, table_path=True,
                        rc_fold=True, pairdot_i8=False):
        """Several rectangular blocks of one matrix in one call (device tensors only).  `blocks` is a
        list of dicts: rows=(lo,hi), cols=(lo,hi), out=<2-D tensor [rows, cols]>, optional
        mirror=<2-D tensor [cols, rows]>, optional triangular=True (rows == cols)."""
        import torch
        f32 = str(dtype) in ("float32", "torch.float32", "f32")
        want = torch.float32 if f32 else torch.float64
        arr = (PoBlock * max(1, len(blocks)))()
        for k, b in zip(arr, blocks):
            (k.row_begin, k.row_end), (k.col_begin, k.col_end) = b["rows"], b["cols"]
            out = b["out"]
            assert out.dtype == want and out.is_cuda and out.dim() == 2 and out.stride(1) == 1
            assert out.shape[0] >= k.row_end - k.row_begin and out.shape[1] >= k.col_end - k.col_begin
            k.out, k.ld_out = out.data_ptr(), out.stride(0)
            m = b.get("mirror")
            if m is not None:
                assert m.dtype == want and m.is_cuda and m.dim() == 2 and m.stride(1) == 1
                assert m.shape[0] >= k.col_end - k.col_begin and m.shape[1] >= k.row_end - k.row_begin
                k.mirror, k.ld_mirror = m.data_ptr(), m.stride(0)
            k.triangular = 1 if b.get("triangular") else 0
        n, dim = counts.shape
        stats = PoStats()
        self._use_torch_stream()
        check(self._lib.po_pairwise_blocks_dev(self._h, counts.data_ptr(), totals.data_ptr(), n, dim, METRICS[metric],
                                               PO_F32 if f32 else PO_F64, arr, len(blocks),
                                               (0 if table_path else PO_FLAG_NO_TABLE_PATH) |
                                               (0 if rc_fold else PO_FLAG_NO_RC_FOLD) |
                                               (PO_FLAG_PAIRDOT_I8 if pairdot_i8 else 0),
                                               ctypes.byref(stats) if want_stats else None))
        if want_stats:
            return {"prep_ms": stats.prep_ms, "kernel_ms": stats.kernel_ms, "total_ms": stats.total_ms,
                    "pairs": stats.pairs, "tiles": stats.tiles, "kernel_id": stats.kernel_id,
                    "rc_folded": bool(stats.rc_folded)}
        return None

    def _pairwise(self, counts, totals, freq, metric, row_begin, row_end, dtype, symmetric, out, want_stats,
                  extra_flags=0):
        if metric not in METRICS:
            raise _lib.PhyloligoError(_lib.PO_EINVAL, "unknown metric %r" % (metric,))
        src = freq if freq is not None else counts
        n, dim = src.shape
        row_end = n if row_end is None else row_end
        rows = max(0, row_end - row_begin)
        f32 = str(dtype) in ("float32", "torch.float32", "f32")
        code = PO_F32 if f32 else PO_F64
        flags = (0 if symmetric else PO_FLAG_NO_SYMMETRY) | extra_flags
        stats = PoStats()
        sp = ctypes.byref(stats) if want_stats else None
        if _is_torch(src):
            import torch
            want = torch.float32 if f32 else torch.float64
            if src.device.type != "cuda" or src.device.index != self.device:
                raise _lib.PhyloligoError(_lib.PO_EINVAL, "input tensor is on %s, this context drives cuda:%d" % (src.device, self.device))
            if out is None:
                out = torch.empty((rows, n), dtype=want, device=src.device)
            elif not (_is_torch(out) and out.dtype == want and out.device == src.device and out.dim() == 2 and
                      out.shape[0] >= rows and out.shape[1] >= n and (out.stride(1) == 1 or out.shape[1] <= 1)):
                raise _lib.PhyloligoError(_lib.PO_EINVAL, "out must be a %s tensor [>=%d, >=%d] with unit inner stride on %s"
                                          % (want, rows, n, src.device))
            ld = out.stride(0) if rows > 1 else max(n, out.stride(0) if out.dim() == 2 else n)
            self._use_torch_stream()
            if freq is not None:
                assert freq.dtype == torch.float64 and freq.is_contiguous()
                check(self._lib.po_pairwise_freq_dev(self._h, freq.data_ptr(), n, dim, METRICS[metric], row_begin,
                                                     row_end, code, out.data_ptr(), ld, flags, sp))
            else:
                assert counts.dtype == torch.int32 and totals.dtype == torch.int64
                assert counts.is_contiguous() and totals.is_contiguous()
                check(self._lib.po_pairwise_dev(self._h, counts.data_ptr(), totals.data_ptr(), n, dim, METRICS[metric],
                                                row_begin, row_end, code, out.data_ptr(), ld, flags, sp))
        else:
            want = np.float32 if f32 else np.float64
            if out is None:
                out = np.zeros((rows, n), dtype=want)
            elif not (isinstance(out, np.ndarray) and out.dtype == want and out.ndim == 2 and out.flags.writeable and
                      out.shape[0] >= rows and out.shape[1] >= n and
                      (out.strides[1] == out.itemsize or out.shape[1] <= 1) and out.strides[0] % out.itemsize == 0 and
                      (out.strides[0] >= n * out.itemsize or rows <= 1)):
                raise _lib.PhyloligoError(_lib.PO_EINVAL, "out must be a writeable %s array [>=%d, >=%d] with unit inner stride"
                                          % (np.dtype(want).name, rows, n))
            ld = out.strides[0] // out.itemsize if rows > 0 and n > 0 else n
            if freq is not None:
                freq = np.ascontiguousarray(freq, dtype=np.float64)
                check(self._lib.po_pairwise_freq(self._h, _np_ptr(freq), n, dim, METRICS[metric], row_begin, row_end,
                                                 code, _np_ptr(out), max(ld, n), flags, sp))
            else:
                counts = np.ascontiguousarray(counts, dtype=np.uint32)
                totals = np.ascontiguousarray(totals, dtype=np.uint64)
                check(self._lib.po_pairwise(self._h, _np_ptr(counts), _np_ptr(totals), n, dim, METRICS[metric],
                                            row_begin, row_end, code, _np_ptr(out), max(ld, n), flags, sp))
        if want_stats:
            return out, {"prep_ms": stats.prep_ms, "kernel_ms": stats.kernel_ms, "total_ms": stats.total_ms,
                         "pairs": stats.pairs, "tiles": stats.tiles, "kernel_id": stats.kernel_id,
                    "rc_folded": bool(stats.rc_folded)}
        return out


# ---- host formats -------------------------------------------------------------------------------
class _Titles:
    """Record titles of a FASTA file, decoded on demand (50 000 Python strings cost more than the parse)."""

    def __init__(self, blob, bounds):
        self._blob, self._bounds = blob, bounds

    def __len__(self):
        return len(self._bounds) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._blob[int(self._bounds[i]):int(self._bounds[i + 1])].tobytes().decode("latin-1")

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __eq__(self, other):
        return list(self) == list(other)


def fasta_index(data):
    """bytes / numpy uint8 of a FASTA file -> (seq uint8[total], offsets uint64[n+1], titles sequence)."""
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    buf = np.ascontiguousarray(buf)
    nrec, nbytes = ctypes.c_uint64(), ctypes.c_uint64()
    check(lib.po_fasta_scan(_np_ptr(buf), buf.shape[0], ctypes.byref(nrec), ctypes.byref(nbytes)))
    seq = np.empty(nbytes.value, dtype=np.uint8)
    offsets = np.zeros(nrec.value + 1, dtype=np.uint64)
    tb = np.zeros(nrec.value, dtype=np.uint64)
    te = np.zeros(nrec.value, dtype=np.uint64)
    check(lib.po_fasta_extract(_np_ptr(buf), buf.shape[0], _np_ptr(seq), _np_ptr(offsets), _np_ptr(tb), _np_ptr(te)))
    # title bytes gathered into one blob (the file buffer may be a memmap that goes away)
    lens = (te - tb).astype(np.int64)
    bounds = np.zeros(nrec.value + 1, dtype=np.int64)
    np.cumsum(lens, out=bounds[1:])
    idx = np.repeat(tb.astype(np.int64) - bounds[:-1], lens) + np.arange(int(bounds[-1]), dtype=np.int64)
    return seq, offsets, _Titles(np.asarray(buf[idx]) if idx.size else np.zeros(0, np.uint8), bounds)


class _LineTitles:
    """Record titles of a device-parsed FASTA file: spans [begin, line end) into the host copy of the file bytes,
    right-stripped when decoded (the device leaves the trailing white space to whoever reads a title)."""

    def __init__(self, buf, begins, ends):
        self._buf, self._b, self._e = buf, begins, ends

    def __len__(self):
        return len(self._b)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._buf[int(self._b[i]):int(self._e[i])].tobytes().rstrip(b" \t\n\r\x0b\x0c").decode("latin-1")

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __eq__(self, other):
        return list(self) == list(other)


def fasta_index_dev(ctx, path, phases=None):
    """FASTA file -> (seq uint8 CUDA tensor, offsets int64 CUDA tensor [n+1], titles): the file bytes go to HBM as they
    are and are parsed there (po_fasta_scan_dev / po_fasta_extract_dev).  Raises PhyloligoError(PO_EUNSUPPORTED) for the
    one construct left to the host parser (tabs on sequence lines): callers fall back to fasta_index().
    phases: a dict that receives the wall time of every step in ms (the device is synchronised between steps only then)."""
    import os
    import time
    import torch
    lib = _lib.load()
    size = os.path.getsize(path)
    dev = torch.device("cuda", ctx.device)

    t_last = [time.perf_counter()]

    def mark(name):
        if phases is not None:
            torch.cuda.synchronize(dev)
            now = time.perf_counter()
            phases[name] = phases.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now

    buf = np.empty(size, dtype=np.uint8)
    check(lib.po_file_read(str(path).encode(), _np_ptr(buf), size))
    mark("file_read_ms")
    raw = torch.from_numpy(buf).to(dev)
    mark("file_h2d_ms")
    nrec, nbytes = ctypes.c_uint64(), ctypes.c_uint64()
    ctx._use_torch_stream()
    check(lib.po_fasta_scan_dev(ctx._h, raw.data_ptr(), size, ctypes.byref(nrec), ctypes.byref(nbytes)))
    mark("fasta_scan_ms")
    seq = torch.empty(((nbytes.value + 15) // 16 * 16 or 16,), dtype=torch.uint8, device=dev)
    offsets = torch.zeros((nrec.value + 1,), dtype=torch.int64, device=dev)
    tb = torch.empty((max(1, nrec.value),), dtype=torch.int64, device=dev)
    te = torch.empty((max(1, nrec.value),), dtype=torch.int64, device=dev)
    mark("device_alloc_ms")
    check(lib.po_fasta_extract_dev(ctx._h, raw.data_ptr(), size, seq.data_ptr(), offsets.data_ptr(), tb.data_ptr(), te.data_ptr()))
    mark("fasta_extract_ms")
    titles = _LineTitles(buf, tb[:nrec.value].cpu().numpy(), te[:nrec.value].cpu().numpy())
    mark("title_spans_d2h_ms")
    return seq[:nbytes.value], offsets, titles


def write_mat_text(path, m, append=False):
    """numpy.savetxt(path, m, delimiter="\\t") byte for byte (bin/phyloligo.py:1061,1066)."""
    lib = _lib.load()
    if m is None:
        raise _lib.PhyloligoError(_lib.PO_EINVAL, "write_mat_text: no matrix to write (got None)")
    m = np.ascontiguousarray(m, dtype=np.float64)
    if m.ndim == 1:
        m = m.reshape(-1, 1)      # savetxt writes a 1-D array one value per line
    rows, cols = m.shape
    check(lib.po_write_mat_text(_np_ptr(m), rows, cols, cols, str(path).encode(), 1 if append else 0))

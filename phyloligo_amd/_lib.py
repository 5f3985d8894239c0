"""ctypes binding of libphyloligo_amd.so (include/phyloligo_amd.h).

The shared library is the product: there is no Python or CPU fallback.  If it has not been
built (`python -c "import __graft_entry__ as g; g.build()"` or `make -C phyloligo_amd/csrc`)
importing this module raises, and every compute call fails with PO_ENODEV when no HIP device
is visible.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libphyloligo_amd.so")
# Experiments only (tools/exp/ab.sh): load a variant build from somewhere else instead of overwriting the product
# library in place.  The name starts with PO_ so that bench.py records it among config.env_knobs.
# A variable left over in a shell must not silently swap the library under tests or benchmarks: it is honoured
# only together with the explicit opt-in PO_ALLOW_VARIANT=1 (which ab.sh sets); without it the import fails loudly.
if os.environ.get("PO_LIB_PATH"):
    if os.environ.get("PO_ALLOW_VARIANT") != "1":
        raise ImportError("phyloligo_amd: PO_LIB_PATH=%s is set without PO_ALLOW_VARIANT=1 -- refusing to load a variant "
                          "library in place of the product library (unset PO_LIB_PATH, or opt in for an experiment)"
                          % os.environ["PO_LIB_PATH"])
    LIB_PATH = os.path.abspath(os.environ["PO_LIB_PATH"])
    sys.stderr.write("phyloligo_amd: PO_LIB_PATH set -- loading the VARIANT library %s\n" % LIB_PATH)

PO_OK, PO_EINVAL, PO_ENODEV, PO_ENOMEM, PO_EHIP, PO_EUNSUPPORTED, PO_EIO = 0, -1, -2, -3, -4, -5, -6
STRANDS = {"both": 0, "plus": 1, "minus": 2}
METRICS = {"Eucl": 0, "JSD": 1, "KT": 2, "BC": 3, "SC": 4}
PO_KL = 5          # Kount.py only (po_profile_distances)
PO_F64, PO_F32 = 0, 1
PO_FLAG_NO_SYMMETRY = 1
PO_FLAG_NO_TABLE_PATH = 2
PO_FLAG_NO_RC_FOLD = 4
PO_FLAG_PAIRDOT_I8 = 8
PO_FLAG_NO_PAIRDOT = 16


class PoStats(ctypes.Structure):
    _fields_ = [("prep_ms", ctypes.c_double), ("kernel_ms", ctypes.c_double), ("total_ms", ctypes.c_double),
                ("pairs", ctypes.c_uint64), ("tiles", ctypes.c_uint64), ("kernel_id", ctypes.c_uint32),
                ("rc_folded", ctypes.c_uint32)]


class PoBlock(ctypes.Structure):
    _fields_ = [("row_begin", ctypes.c_uint64), ("row_end", ctypes.c_uint64), ("col_begin", ctypes.c_uint64),
                ("col_end", ctypes.c_uint64), ("out", ctypes.c_void_p), ("ld_out", ctypes.c_uint64),
                ("mirror", ctypes.c_void_p), ("ld_mirror", ctypes.c_uint64), ("triangular", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


class PhyloligoError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("phyloligo_amd: %s (status %d)" % (message, status))
        self.status = status


_c = ctypes
_vp, _u64, _u32, _int, _cp = _c.c_void_p, _c.c_uint64, _c.c_uint32, _c.c_int, _c.c_char_p

# name -> (restype, argtypes); every symbol include/phyloligo_amd.h declares
SIGNATURES = {
    "po_version": (_cp, []),
    "po_abi_version": (_int, []),
    "po_last_error": (_cp, []),
    "po_status_string": (_cp, [_int]),
    "po_device_count": (_int, []),
    "po_ctx_create": (_int, [_c.POINTER(_vp), _int]),
    "po_ctx_destroy": (None, [_vp]),
    "po_ctx_set_stream": (_int, [_vp, _vp]),
    "po_ctx_synchronize": (_int, [_vp]),
    "po_ctx_device_name": (_int, [_vp, _c.c_char_p, _c.c_size_t]),
    "po_ctx_trim": (_int, [_vp]),
    "po_pattern_info": (_int, [_cp, _c.POINTER(_u32), _c.POINTER(_u32), _c.POINTER(_u64)]),
    "po_count_profiles": (_int, [_vp, _vp, _vp, _u64, _cp, _int, _vp, _vp]),
    "po_count_profiles_dev": (_int, [_vp, _vp, _vp, _u64, _u64, _cp, _int, _vp, _vp]),
    "po_count_profiles_ranges": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _cp, _int, _vp, _vp]),
    "po_count_profiles_ranges_dev": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _u64, _cp, _int, _vp, _vp]),
    "po_profile_distances": (_int, [_vp, _vp, _vp, _u64, _u32, _vp, _int, _vp]),
    "po_profile_distances_dev": (_int, [_vp, _vp, _vp, _u64, _u32, _vp, _int, _vp]),
    "po_count_byte_ranges_dev": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _int, _vp]),
    "po_frequencies": (_int, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "po_frequencies_dev": (_int, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "po_pairwise": (_int, [_vp, _vp, _vp, _u64, _u32, _int, _u64, _u64, _int, _vp, _u64, _u32, _c.POINTER(PoStats)]),
    "po_pairwise_dev": (_int, [_vp, _vp, _vp, _u64, _u32, _int, _u64, _u64, _int, _vp, _u64, _u32, _c.POINTER(PoStats)]),
    "po_pairwise_freq": (_int, [_vp, _vp, _u64, _u32, _int, _u64, _u64, _int, _vp, _u64, _u32, _c.POINTER(PoStats)]),
    "po_pairwise_freq_dev": (_int, [_vp, _vp, _u64, _u32, _int, _u64, _u64, _int, _vp, _u64, _u32, _c.POINTER(PoStats)]),
    "po_pairwise_blocks_dev": (_int, [_vp, _vp, _vp, _u64, _u32, _int, _int, _c.POINTER(PoBlock), _u32, _u32, _c.POINTER(PoStats)]),
    "po_pairwise_reserve": (_int, [_vp, _u64, _u32, _int]),
    "po_fasta_scan": (_int, [_vp, _u64, _c.POINTER(_u64), _c.POINTER(_u64)]),
    "po_fasta_extract": (_int, [_vp, _u64, _vp, _vp, _vp, _vp]),
    "po_file_read": (_int, [_cp, _vp, _u64]),
    "po_fasta_scan_dev": (_int, [_vp, _vp, _u64, _c.POINTER(_u64), _c.POINTER(_u64)]),
    "po_fasta_extract_dev": (_int, [_vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    "po_write_mat_text": (_int, [_vp, _u64, _u64, _u64, _cp, _int]),
    "po_pwrite_rows": (_int, [_int, _vp, _u64, _u64, _u64, _u64, _u64, _int]),
}

_lib = None
# Set by `python -m phyloligo_amd` (single process) before the first load(): the CLI needs no torch - host-pointer entry
# points do the copies - and importing it costs ~0.7 s of a 2 - 3 s run.  Only for a process that will never import torch
# afterwards: torch bundles its own libamdhip64, and a process must hold exactly one HIP runtime.
PREFER_NO_TORCH = False


def load():
    """Load the library once.  torch (when installed) is imported first: it bundles its own
    libamdhip64.so.7, and a process must hold exactly one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "phyloligo_amd: %s is missing -- build it (make -C phyloligo_amd/csrc, needs hipcc). "
            "There is no CPU fallback." % LIB_PATH)
    if "torch" not in sys.modules and not PREFER_NO_TORCH:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header and library out of step
        fn.restype = res
        fn.argtypes = args
    if lib.po_abi_version() != 1:
        raise ImportError("phyloligo_amd: ABI version mismatch")
    _lib = lib
    return lib


def check(status):
    if status != PO_OK:
        lib = load()
        msg = lib.po_last_error().decode("utf-8", "replace") or lib.po_status_string(status).decode()
        raise PhyloligoError(status, msg)

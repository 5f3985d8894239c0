"""The HDF5 container of `--large h5py` without h5py: a few calls into libhdf5 through ctypes.

Reference: /root/reference/phylopackage/bin/phyloligo.py:456-478 (`join_distance_results`: one file, one dataset
`"distances"`, shape (N, N), dtype float32 - h5py's defaults: contiguous layout, little endian) and :918-929 (`"frequencies"`,
(N, 4^k), float32); read back by phyloselect.py:615-619 (`hf.get("distances")`) and phyloligo_comparemat.py:10-14.

h5py is not in this image, but libhdf5 is (1.10.6 under /opt/conda/lib, with h5dump / h5ls).  The dataset is created with EARLY
allocation and no fill, so that after the file is closed its raw data is one contiguous byte range at a known offset
(`H5Dget_offset`) - which the float32 writer of the memmap variant then fills with parallel pwrite (`po_pwrite_rows`), from one
process or from one rank per GPU, exactly as it fills the raw container.  Nothing here touches a GPU.
"""
import ctypes
import ctypes.util
import os

_lib = None
_CANDIDATES = ("libhdf5.so", "/opt/conda/lib/libhdf5.so", "libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so")

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5D_ALLOC_TIME_EARLY, H5D_FILL_TIME_NEVER = 1, 1
HADDR_UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Unavailable(RuntimeError):
    pass


def _load():
    """libhdf5 >= 1.10 (64-bit hid_t) or Hdf5Unavailable"""
    global _lib
    if _lib is not None:
        return _lib
    names = [os.environ["PO_HDF5_LIB"]] if os.environ.get("PO_HDF5_LIB") else []
    found = ctypes.util.find_library("hdf5")
    names += ([found] if found else []) + list(_CANDIDATES)
    lib = None
    for name in names:
        try:
            lib = ctypes.CDLL(name)
            break
        except OSError:
            continue
    if lib is None:
        raise Hdf5Unavailable("no libhdf5 found (tried %s)" % ", ".join(names))
    i64, u64, cint, vp, cp = ctypes.c_int64, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p
    sig = {"H5open": (cint, []), "H5get_libversion": (cint, [ctypes.POINTER(ctypes.c_uint)] * 3),
           "H5Fcreate": (i64, [cp, ctypes.c_uint, i64, i64]), "H5Fopen": (i64, [cp, ctypes.c_uint, i64]), "H5Fclose": (cint, [i64]),
           "H5Screate_simple": (i64, [cint, ctypes.POINTER(u64), ctypes.POINTER(u64)]), "H5Sclose": (cint, [i64]),
           "H5Sget_simple_extent_ndims": (cint, [i64]),
           "H5Sget_simple_extent_dims": (cint, [i64, ctypes.POINTER(u64), ctypes.POINTER(u64)]),
           "H5Pcreate": (i64, [i64]), "H5Pclose": (cint, [i64]), "H5Pset_alloc_time": (cint, [i64, cint]),
           "H5Pset_fill_time": (cint, [i64, cint]),
           "H5Dcreate2": (i64, [i64, cp, i64, i64, i64, i64, i64]), "H5Dopen2": (i64, [i64, cp, i64]), "H5Dclose": (cint, [i64]),
           "H5Dget_offset": (u64, [i64]), "H5Dget_space": (i64, [i64]), "H5Dget_type": (i64, [i64]),
           "H5Dread": (cint, [i64, i64, i64, i64, i64, vp]), "H5Tequal": (cint, [i64, i64]), "H5Tclose": (cint, [i64])}
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise Hdf5Unavailable("H5open failed")
    maj, mi, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mi), ctypes.byref(rel))
    if (maj.value, mi.value) < (1, 10):
        raise Hdf5Unavailable("libhdf5 %d.%d.%d: hid_t is 32 bits before 1.10" % (maj.value, mi.value, rel.value))
    lib.po_version = "%d.%d.%d" % (maj.value, mi.value, rel.value)
    _lib = lib
    return lib


def available():
    try:
        _load()
        return True
    except Hdf5Unavailable:
        return False


def _gid(lib, name):
    """a library global of type hid_t (H5T_IEEE_F32LE_g, H5P_CLS_DATASET_CREATE_ID_g ...), valid after H5open()"""
    return ctypes.c_int64.in_dll(lib, name).value


def _check(value, what):
    if value < 0:
        raise OSError("libhdf5: %s failed" % what)
    return value


def create_f32_dataset(path, name, rows, cols):
    """New file `path` with ONE dataset `name` of shape (rows, cols), float32 little endian, contiguous, space allocated, nothing
    written.  Returns the byte offset of its raw data in the (closed) file: element (i, j) lives at offset + 4 (i cols + j)."""
    lib = _load()
    f = _check(lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), "H5Fcreate(%s)" % path)
    try:
        dims = (ctypes.c_uint64 * 2)(int(rows), int(cols))
        space = _check(lib.H5Screate_simple(2, dims, None), "H5Screate_simple")
        dcpl = _check(lib.H5Pcreate(_gid(lib, "H5P_CLS_DATASET_CREATE_ID_g")), "H5Pcreate")
        try:
            _check(lib.H5Pset_alloc_time(dcpl, H5D_ALLOC_TIME_EARLY), "H5Pset_alloc_time")
            _check(lib.H5Pset_fill_time(dcpl, H5D_FILL_TIME_NEVER), "H5Pset_fill_time")
            d = _check(lib.H5Dcreate2(f, name.encode(), _gid(lib, "H5T_IEEE_F32LE_g"), space, H5P_DEFAULT, dcpl, H5P_DEFAULT),
                       "H5Dcreate2(%s)" % name)
            try:
                offset = lib.H5Dget_offset(d)
            finally:
                lib.H5Dclose(d)
        finally:
            lib.H5Pclose(dcpl)
            lib.H5Sclose(space)
    finally:
        _check(lib.H5Fclose(f), "H5Fclose")
    if rows * cols and offset == HADDR_UNDEF:
        raise OSError("libhdf5: the dataset has no address (not contiguous?)")
    need = (0 if offset == HADDR_UNDEF else offset) + rows * cols * 4
    if os.path.getsize(path) < need:                 # the library sets the file size to its end of allocation; make sure of it
        with open(path, "r+b") as fh:
            fh.truncate(need)
    return 0 if offset == HADDR_UNDEF else int(offset)


def read_f32_dataset(path, name):
    """The dataset as a float32 numpy array, read by libhdf5 itself (what `hf.get(name).value[:]` of the reference's readers does)."""
    import numpy as np
    lib = _load()
    f = _check(lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT), "H5Fopen(%s)" % path)
    try:
        d = _check(lib.H5Dopen2(f, name.encode(), H5P_DEFAULT), "H5Dopen2(%s)" % name)
        try:
            space = _check(lib.H5Dget_space(d), "H5Dget_space")
            nd = lib.H5Sget_simple_extent_ndims(space)
            dims = (ctypes.c_uint64 * max(1, nd))()
            lib.H5Sget_simple_extent_dims(space, dims, None)
            lib.H5Sclose(space)
            t = _check(lib.H5Dget_type(d), "H5Dget_type")
            is_f32 = lib.H5Tequal(t, _gid(lib, "H5T_IEEE_F32LE_g")) > 0
            lib.H5Tclose(t)
            if not is_f32:
                raise OSError("dataset %s of %s is not float32 little endian" % (name, path))
            out = np.empty(tuple(int(x) for x in dims[:nd]), dtype=np.float32)
            if out.size:
                _check(lib.H5Dread(d, _gid(lib, "H5T_NATIVE_FLOAT_g"), H5S_ALL, H5S_ALL, H5P_DEFAULT, ctypes.c_void_p(out.ctypes.data)),
                       "H5Dread")
            return out
        finally:
            lib.H5Dclose(d)
    finally:
        lib.H5Fclose(f)

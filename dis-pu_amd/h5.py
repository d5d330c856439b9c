"""Read-only HDF5 access for the training-patch file (the data format on the input side of the path).

The reference opens `PUGAN_poisson_256_poisson_1024.h5` with h5py and slices whole datasets
(DisPU/dataset.py:52-78: `f['poisson_%d' % num][:]`).  h5py is not in this image, but the HDF5 C library it wraps is
(`libhdf5.so`, 1.10.x, under /opt/conda/lib), so the same library is bound directly with ctypes: open file, open
dataset, query the dataspace and datatype, H5Dread into a numpy array of the file's own type.  Every storage layout and
filter the library supports (contiguous, chunked, gzip, shuffle) therefore reads exactly as it does through h5py.

Library lookup order: $DISPU_HDF5_LIB, ctypes.util.find_library('hdf5'), /opt/conda/lib, the usual system directories.
A missing library is an error (no fallback format guessing).
"""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

_LIB = None
_hid_t = ctypes.c_int64                      # HDF5 >= 1.10
_hsize_t = ctypes.c_uint64
_herr_t = ctypes.c_int

H5F_ACC_RDONLY = 0
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT = 0, 1                # H5T_class_t
H5T_ORDER_LE, H5T_ORDER_BE = 0, 1
H5T_SGN_NONE, H5T_SGN_2 = 0, 1
H5T_DIR_ASCEND = 1
H5O_TYPE_GROUP, H5O_TYPE_DATASET = 0, 1


def _candidates():
    env = os.environ.get("DISPU_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        for path in sorted(glob.glob(pat)):
            base = os.path.basename(path)
            if base.startswith("libhdf5.so") or base.startswith("libhdf5_serial.so"):
                yield path


def lib():
    """The loaded libhdf5 (ctypes.CDLL) with argument / result types declared; raises RuntimeError if there is none."""
    global _LIB
    if _LIB is not None:
        return _LIB
    tried, h = [], None
    for cand in _candidates():
        try:
            h = ctypes.CDLL(cand)
            break
        except OSError as e:
            tried.append("%s (%s)" % (cand, e))
    if h is None:
        raise RuntimeError("HDF5 C library not found; set DISPU_HDF5_LIB to libhdf5.so.  Tried: %s" % (tried or "nothing on the search paths"))
    sig = {
        "H5open": (_herr_t, []),
        "H5get_libversion": (_herr_t, [ctypes.POINTER(ctypes.c_uint)] * 3),
        "H5Fopen": (_hid_t, [ctypes.c_char_p, ctypes.c_uint, _hid_t]),
        "H5Fclose": (_herr_t, [_hid_t]),
        "H5Dopen2": (_hid_t, [_hid_t, ctypes.c_char_p, _hid_t]),
        "H5Dclose": (_herr_t, [_hid_t]),
        "H5Dget_space": (_hid_t, [_hid_t]),
        "H5Dget_type": (_hid_t, [_hid_t]),
        "H5Dread": (_herr_t, [_hid_t, _hid_t, _hid_t, _hid_t, _hid_t, ctypes.c_void_p]),
        "H5Sclose": (_herr_t, [_hid_t]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [_hid_t]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [_hid_t, ctypes.POINTER(_hsize_t), ctypes.POINTER(_hsize_t)]),
        "H5Tclose": (_herr_t, [_hid_t]),
        "H5Tget_class": (ctypes.c_int, [_hid_t]),
        "H5Tget_size": (ctypes.c_size_t, [_hid_t]),
        "H5Tget_order": (ctypes.c_int, [_hid_t]),
        "H5Tget_sign": (ctypes.c_int, [_hid_t]),
        "H5Tget_native_type": (_hid_t, [_hid_t, ctypes.c_int]),
        "H5Gget_num_objs": (_herr_t, [_hid_t, ctypes.POINTER(_hsize_t)]),
        "H5Gget_objname_by_idx": (ctypes.c_ssize_t, [_hid_t, _hsize_t, ctypes.c_char_p, ctypes.c_size_t]),
        "H5Gget_objtype_by_idx": (ctypes.c_int, [_hid_t, _hsize_t]),
        "H5Gopen2": (_hid_t, [_hid_t, ctypes.c_char_p, _hid_t]),
        "H5Gclose": (_herr_t, [_hid_t]),
        "H5Eset_auto2": (_herr_t, [_hid_t, ctypes.c_void_p, ctypes.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    if h.H5open() < 0:
        raise RuntimeError("H5open failed")
    maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    h.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise RuntimeError("libhdf5 %d.%d.%d: this binding assumes the 64-bit hid_t of HDF5 >= 1.10" % (maj.value, mnr.value, rel.value))
    h.H5Eset_auto2(0, None, None)            # errors are reported through return codes -> python exceptions, not stderr dumps
    h.version = (maj.value, mnr.value, rel.value)
    _LIB = h
    return h


def _check(v, what):
    if v < 0:
        raise IOError("HDF5: %s failed" % what)
    return v


def _numpy_dtype(h, tid, name):
    cls, size = h.H5Tget_class(tid), h.H5Tget_size(tid)
    if cls == H5T_FLOAT and size in (2, 4, 8):
        code = "f%d" % size
    elif cls == H5T_INTEGER and size in (1, 2, 4, 8):
        code = ("i%d" if h.H5Tget_sign(tid) == H5T_SGN_2 else "u%d") % size
    else:
        raise TypeError("dataset %r: HDF5 type class %d of %d bytes is not a plain integer / float array" % (name, cls, size))
    return np.dtype(code)                    # read through the NATIVE memory type: the library converts the byte order


class File(object):
    """`with File(path) as f: a = f['poisson_1024']` -- whole-dataset reads, like the reference's `f[name][:]`."""

    def __init__(self, path):
        self._h = lib()
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        self._fid = self._h.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
        if self._fid < 0:
            self._fid = None
            raise IOError("%s is not an HDF5 file the library can open" % path)
        self.path = path

    def close(self):
        if self._fid is not None:
            self._h.H5Fclose(self._fid)
            self._fid = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:                    # noqa: BLE001 -- interpreter shutdown
            pass

    def keys(self, group="/"):
        """Names of the links directly under `group` (datasets and sub-groups, in the library's index order)."""
        h = self._h
        gid = _check(h.H5Gopen2(self._fid, group.encode(), H5P_DEFAULT), "open group %r" % group)
        try:
            n = _hsize_t()
            _check(h.H5Gget_num_objs(gid, ctypes.byref(n)), "count objects")
            out = []
            for i in range(n.value):
                ln = _check(h.H5Gget_objname_by_idx(gid, i, None, 0), "object name length")
                buf = ctypes.create_string_buffer(ln + 1)
                h.H5Gget_objname_by_idx(gid, i, buf, ln + 1)
                out.append(buf.value.decode())
            return out
        finally:
            h.H5Gclose(gid)

    def __contains__(self, name):
        did = self._h.H5Dopen2(self._fid, name.encode(), H5P_DEFAULT)
        if did < 0:
            return False
        self._h.H5Dclose(did)
        return True

    def shape_dtype(self, name):
        h = self._h
        did = h.H5Dopen2(self._fid, name.encode(), H5P_DEFAULT)
        if did < 0:
            raise KeyError("no dataset %r in %s (has: %s)" % (name, self.path, ", ".join(self.keys())))
        try:
            sid = _check(h.H5Dget_space(did), "dataspace of %r" % name)
            tid = _check(h.H5Dget_type(did), "datatype of %r" % name)
            try:
                nd = _check(h.H5Sget_simple_extent_ndims(sid), "rank of %r" % name)
                dims = (_hsize_t * max(nd, 1))()
                if nd:
                    _check(h.H5Sget_simple_extent_dims(sid, dims, None), "extent of %r" % name)
                return tuple(int(dims[i]) for i in range(nd)), _numpy_dtype(h, tid, name)
            finally:
                h.H5Tclose(tid)
                h.H5Sclose(sid)
        finally:
            h.H5Dclose(did)

    def __getitem__(self, name):
        """The whole dataset as a C-contiguous numpy array of the file's element type (native byte order)."""
        h = self._h
        shape, dt = self.shape_dtype(name)
        out = np.empty(shape, dt)
        did = _check(h.H5Dopen2(self._fid, name.encode(), H5P_DEFAULT), "open %r" % name)
        try:
            tid = _check(h.H5Dget_type(did), "datatype of %r" % name)
            mem = _check(h.H5Tget_native_type(tid, H5T_DIR_ASCEND), "native type of %r" % name)
            try:
                if out.size:
                    _check(h.H5Dread(did, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p)), "read %r" % name)
            finally:
                h.H5Tclose(mem)
                h.H5Tclose(tid)
        finally:
            h.H5Dclose(did)
        return out

"""ctypes access to the REFERENCE's own CPU functions (oracle/_ref/*.so, built by oracle/build.py
from the sources under /root/reference).  Test infrastructure only; used to pin the oracle and
to generate tests/golden/*.npz.  `available()` is False when the libraries were never built
(they are git-ignored; the GPU box receives the prebuilt files with the snapshot).

Signatures follow the reference:
  approxmatch_cpu / matchcost_cpu / matchcostgrad_cpu   tf_ops/approxmatch/tf_approxmatch.cpp:23-140
  threenn_cpu / threeinterpolate(_grad)_cpu             tf_ops/interpolation/tf_interpolate.cpp:60-153
  nnsearch                                              tf_ops/nn_distance/tf_nndistance.cpp:21-43
  query_ball_point_cpu / group_point(_grad)_cpu         tf_ops/grouping/query_ball_point.cpp:19-84
  cpp_knn_batch(_omp)                                   libs/nearest_neighbors/knn_.cxx:72-135
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CPU = os.path.join(_HERE, "_ref", "libref_cpu.so")
_KNN = os.path.join(_HERE, "_ref", "libref_knn.so")
_libs = {}


def available():
    return os.path.exists(_CPU) and os.path.exists(_KNN)


def _lib(path):
    if path not in _libs:
        _libs[path] = C.CDLL(path)
    return _libs[path]


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def nnsearch(xyz1, xyz2):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d, i = np.empty((b, n), np.float32), np.empty((b, n), np.int32)
    _lib(_CPU).nnsearch(b, n, m, _p(xyz1), _p(xyz2), _p(d), _p(i))
    return d, i


def threenn(xyz1, xyz2):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d, i = np.empty((b, n, 3), np.float32), np.empty((b, n, 3), np.int32)
    _lib(_CPU).threenn_cpu(b, n, m, _p(xyz1), _p(xyz2), _p(d), _p(i))
    return d, i


def threeinterpolate(points, idx, weight):
    points, idx, weight = _f(points), _i(idx), _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    _lib(_CPU).threeinterpolate_cpu(b, m, c, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def threeinterpolate_grad(points, idx, weight, grad_out):
    points, idx, weight, grad_out = _f(points), _i(idx), _f(weight), _f(grad_out)
    b, m, c = points.shape
    n = idx.shape[1]
    g = np.zeros((b, m, c), np.float32)  # the TF op zero-fills before the call (tf_interpolate.cpp:251)
    _lib(_CPU).threeinterpolate_grad_cpu(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


def query_ball_point(radius, nsample, xyz1, xyz2, idx_init=None):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    r = np.array([radius], np.float32)
    idx = np.zeros((b, m, nsample), np.int32) if idx_init is None else _i(idx_init).copy()
    _lib(_CPU).query_ball_point_cpu(b, n, m, _p(r), int(nsample), _p(xyz1), _p(xyz2), _p(idx))
    return idx


def group_point(points, idx):
    points, idx = _f(points), _i(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    _lib(_CPU).group_point_cpu(b, n, c, m, ns, _p(points), _p(idx), _p(out))
    return out


def group_point_grad(points, idx, grad_out):
    points, idx, grad_out = _f(points), _i(idx), _f(grad_out)
    b, n, c = points.shape
    _, m, ns = idx.shape
    g = np.zeros((b, n, c), np.float32)
    _lib(_CPU).group_point_grad_cpu(b, n, c, m, ns, _p(grad_out), _p(idx), _p(g))
    return g


def approxmatch(xyz1, xyz2):
    """Reference CPU variant: 11 levels, double accumulators, writes match[k*m+l] (n-major) into the
    (b,m,n)-shaped buffer (tf_approxmatch.cpp:31,54,75).  Returned here as [b,n,m] (k-major)."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.zeros((b, n, m), np.float32)
    _lib(_CPU).approxmatch_cpu(b, n, m, _p(xyz1), _p(xyz2), _p(match))
    return match


def matchcost(xyz1, xyz2, match_nm):
    """match_nm is [b,n,m] (the CPU functions' own layout)."""
    xyz1, xyz2, match_nm = _f(xyz1), _f(xyz2), _f(match_nm)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = np.empty((b,), np.float32)
    _lib(_CPU).matchcost_cpu(b, n, m, _p(xyz1), _p(xyz2), _p(match_nm), _p(cost))
    return cost


def matchcostgrad(xyz1, xyz2, match_nm):
    xyz1, xyz2, match_nm = _f(xyz1), _f(xyz2), _f(match_nm)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1, g2 = np.zeros((b, n, 3), np.float32), np.zeros((b, m, 3), np.float32)
    _lib(_CPU).matchcostgrad_cpu(b, n, m, _p(xyz1), _p(xyz2), _p(match_nm), _p(g1), _p(g2))
    return g1, g2


def knn_batch(pts, queries, K, omp=False):
    pts, queries = _f(pts), _f(queries)
    b, n, dim = pts.shape
    m = queries.shape[1]
    out = np.zeros((b, m, K), np.int64)
    fn = getattr(_lib(_KNN), "_Z17cpp_knn_batch_ompPKfmmmS0_mmPl" if omp else "_Z13cpp_knn_batchPKfmmmS0_mmPl")
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    fn(_p(pts), b, n, dim, _p(queries), m, K, _p(out))
    return out

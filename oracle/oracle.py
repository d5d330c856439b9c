"""numpy front-end of the CPU oracle (oracle/*.c -> oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), and the
cpu_baseline leg of bench.py.  The product package (dis-pu_amd/) never imports this module.

Function names/argument order mirror the reference's Python op wrappers
(tf_ops/*/tf_*.py, libs/nearest_neighbors/knn.pyx); `contract` selects the pinned arithmetic
documented at the top of dispu_oracle.c (0 = reference CPU flavour, 1 = nvcc-contracted flavour).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            import importlib.util
            spec = importlib.util.spec_from_file_location("_oracle_build", os.path.join(_HERE, "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build_oracle()
        _LIB = C.CDLL(path)
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_threads(t):
    lib().orc_set_threads(int(t))


# ---- A1/A2 sampling (tf_ops/sampling/tf_sampling.py:29-56) -------------------------------------
def farthest_point_sample(npoint, inp, contract=1, block=512):
    inp = _f(inp)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    lib().orc_fps(b, n, int(npoint), _p(inp), _p(out), int(contract), int(block))
    return out


def prob_sample(inp, inpr, return_temp=False):
    """tf_sampling.prob_sample (tf_sampling.py:12-27): inp [b, ncategory] weights, inpr [b, npoints] uniform randoms."""
    inp, inpr = _f(inp), _f(inpr)
    b, n = inp.shape
    m = inpr.shape[1]
    temp = np.empty((b, n), np.float32)
    out = np.empty((b, m), np.int32)
    lib().orc_prob_sample(b, n, m, _p(inp), _p(inpr), _p(temp), _p(out))
    return (out, temp) if return_temp else out


def gather_point(inp, idx):
    inp, idx = _f(inp), _i(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.empty((b, m, 3), np.float32)
    lib().orc_gather_point(b, n, m, _p(inp), _p(idx), _p(out))
    return out


def gather_point_grad(inp, idx, out_g):
    inp, idx, out_g = _f(inp), _i(idx), _f(out_g)
    b, n, _ = inp.shape
    m = idx.shape[1]
    g = np.empty((b, n, 3), np.float32)
    lib().orc_gather_point_grad(b, n, m, _p(out_g), _p(idx), _p(g))
    return g


# ---- A3/A4 grouping (tf_ops/grouping/tf_grouping.py:9-57) ---------------------------------------
def query_ball_point(radius, nsample, xyz, new_xyz, contract=1, idx_init=None):
    """Returns (idx[b,m,ns], pts_cnt[b,m]).  Rows without a hit keep `idx_init` (default 0):
    the reference leaves them unwritten."""
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    r = np.full((b,), radius, np.float32) if np.isscalar(radius) else _f(radius)
    idx = np.zeros((b, m, nsample), np.int32) if idx_init is None else _i(idx_init).copy()
    cnt = np.zeros((b, m), np.int32)
    lib().orc_query_ball(b, n, m, _p(r), int(nsample), _p(xyz), _p(new_xyz), _p(idx), _p(cnt), int(contract))
    return idx, cnt


def group_point(points, idx):
    points, idx = _f(points), _i(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    lib().orc_group_point(b, n, c, m, ns, _p(points), _p(idx), _p(out))
    return out


def group_point_grad(points, idx, grad_out):
    points, idx, grad_out = _f(points), _i(idx), _f(grad_out)
    b, n, c = points.shape
    _, m, ns = idx.shape
    g = np.empty((b, n, c), np.float32)
    lib().orc_group_point_grad(b, n, c, m, ns, _p(grad_out), _p(idx), _p(g))
    return g


def select_top_k(k, dist):
    dist = _f(dist)
    b, m, n = dist.shape
    outi = np.empty((b, m, n), np.int32)
    out = np.empty((b, m, n), np.float32)
    lib().orc_selection_sort(b, n, m, int(k), _p(dist), _p(outi), _p(out))
    return outi, out


def knn_point(k, xyz1, xyz2):
    """(val = -d2 [b,m,k], idx [b,m,k]); xyz1 = dataset [b,n,c], xyz2 = queries [b,m,c]."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    val = np.empty((b, m, k), np.float32)
    idx = np.empty((b, m, k), np.int32)
    lib().orc_knn_point(b, n, m, c, int(k), _p(xyz1), _p(xyz2), _p(val), _p(idx))
    return val, idx


def knn_point_2(k, points, queries, sort=True, unique=True):
    """(dist [b,P,k], indices [b,P,k,2]) like tf_grouping.py:95-114 (batch index, point index)."""
    points, queries = _f(points), _f(queries)
    b, n, c = points.shape
    m = queries.shape[1]
    dist = np.empty((b, m, k), np.float32)
    idx = np.empty((b, m, k), np.int32)
    lib().orc_knn_feat(b, n, m, c, int(k), _p(points), _p(queries), _p(dist), _p(idx))
    bidx = np.broadcast_to(np.arange(b, dtype=np.int32)[:, None, None], idx.shape)
    return dist, np.stack([bidx, idx], axis=-1)


def knn_batch(pts, queries, K, omp=False, contract=0, return_dist=False):
    """nearest_neighbors.knn_batch (knn.pyx:71-109): int64 [B,N2,K]."""
    pts, queries = _f(pts), _f(queries)
    b, n, _ = pts.shape
    m = queries.shape[1]
    idx = np.empty((b, m, K), np.int32)
    dist = np.empty((b, m, K), np.float32)
    lib().orc_knn_xyz(b, n, m, int(K), _p(pts), _p(queries), _p(idx), _p(dist), int(contract))
    if return_dist:
        return idx.astype(np.int64), dist
    return idx.astype(np.int64)


# ---- A8/A9 interpolation (tf_ops/interpolation/tf_interpolate.py:8-34) ----------------------------
def three_nn(xyz1, xyz2, contract=0):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    with np.errstate(over="ignore"):
        lib().orc_three_nn(b, n, m, _p(xyz1), _p(xyz2), _p(dist), _p(idx), int(contract))
    return dist, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f(points), _i(idx), _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    lib().orc_three_interpolate(b, m, c, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(points, idx, weight, grad_out):
    points, idx, weight, grad_out = _f(points), _i(idx), _f(weight), _f(grad_out)
    b, m, c = points.shape
    n = idx.shape[1]
    g = np.empty((b, m, c), np.float32)
    lib().orc_three_interpolate_grad(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


# ---- A10 nn_distance (tf_ops/nn_distance/tf_nndistance.py:14-37) ---------------------------------
def nn_distance(xyz1, xyz2, contract=1):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1, i1 = np.empty((b, n), np.float32), np.empty((b, n), np.int32)
    d2, i2 = np.empty((b, m), np.float32), np.empty((b, m), np.int32)
    lib().orc_nn_distance(b, n, m, _p(xyz1), _p(xyz2), _p(d1), _p(i1), _p(d2), _p(i2), int(contract))
    return d1, i1, d2, i2


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1, g2 = np.empty((b, n, 3), np.float32), np.empty((b, m, 3), np.float32)
    gd1, gd2, idx1, idx2 = _f(grad_dist1), _f(grad_dist2), _i(idx1), _i(idx2)
    lib().orc_nn_distance_grad(b, n, m, _p(xyz1), _p(xyz2), _p(gd1), _p(idx1), _p(gd2), _p(idx2), _p(g1), _p(g2))
    return g1, g2


# ---- A11/A12 approxmatch (tf_ops/approxmatch/tf_approxmatch.py:13-51) ----------------------------
AM_CHUNK = 128      # csrc/approxmatch.hip AM_CH: partners per partial sum of the MI355X kernels


def approx_match(xyz1, xyz2, contract=1, pinned_exp=False, chunk=0):
    """chunk = 0: the reference kernel's sequential summation order; chunk = AM_CHUNK: the MI355X kernels' order
    (partial sums over consecutive pieces of `chunk` partners, added in ascending order)."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.empty((b, m, n), np.float32)
    lib().orc_approx_match_chunked(b, n, m, _p(xyz1), _p(xyz2), _p(match), int(contract), int(bool(pinned_exp)), int(chunk))
    return match


def match_cost(xyz1, xyz2, match, contract=1, block=512):
    xyz1, xyz2, match = _f(xyz1), _f(xyz2), _f(match)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = np.empty((b,), np.float32)
    lib().orc_match_cost(b, n, m, _p(xyz1), _p(xyz2), _p(match), _p(cost), int(contract), int(block))
    return cost


def match_cost_grad(xyz1, xyz2, match, contract=1):
    xyz1, xyz2, match = _f(xyz1), _f(xyz2), _f(match)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1, g2 = np.empty((b, n, 3), np.float32), np.empty((b, m, 3), np.float32)
    lib().orc_match_cost_grad(b, n, m, _p(xyz1), _p(xyz2), _p(match), _p(g1), _p(g2), int(contract))
    return g1, g2

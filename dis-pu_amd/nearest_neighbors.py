"""Batched exact k-NN on xyz -- the interface of the reference's Cython/nanoflann extension
libs/nearest_neighbors/knn.pyx:71-109 (`knn_batch`) and its caller Common/ops.py:110-118
(`knn_query`), computed on the device instead of a host KD-tree round trip."""
import torch

from . import _lib
from ._util import f32, req


def _knn_xyz(b, n, m, k, pts, queries, idx, dist, arith):
    """dispu_knn_xyz_ws with the scratch the chunked path asks for (clouds of 1025 .. 8192 points; 0 bytes otherwise)."""
    L = _lib.lib()
    nbytes = L.dispu_knn_xyz_scratch_bytes(b, n, m, k)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=pts.device) if nbytes else None
    _lib.check(L.dispu_knn_xyz_ws(b, n, m, k, _lib.ptr(pts), _lib.ptr(queries), _lib.ptr(idx), _lib.ptr(dist), _lib.ptr(scratch), nbytes,
                                  arith, _lib.stream_ptr(pts.device)), "dispu_knn_xyz")


def knn_batch(pts, queries, K, omp=False, return_dist=False, arith=_lib.ARITH_PLAIN):
    """(pts[B,N1,3] f32, queries[B,N2,3] f32, K) -> int64 [B,N2,K]  (ascending distance; self first when the
    query is a support point).  `omp` is accepted for signature parity (the reference switches between a
    serial and an OpenMP host loop, knn.pyx:99-107).  Exact ties: lower index first (the reference's
    order on exact ties depends on KD-tree traversal)."""
    pts, queries = f32(pts, "pts"), f32(queries, "queries")
    req(pts.dim() == 3 and pts.shape[2] == 3 and queries.dim() == 3 and queries.shape[2] == 3
        and pts.shape[0] == queries.shape[0], "knn_batch expects pts (B,N1,3) and queries (B,N2,3)")
    b, n, _ = pts.shape
    m = queries.shape[1]
    req(0 < int(K) <= n and int(K) <= 4096, "knn_batch supports 1 <= K <= min(N1, 4096)")
    idx = torch.empty((b, m, int(K)), dtype=torch.int32, device=pts.device)
    dist = torch.empty((b, m, int(K)), dtype=torch.float32, device=pts.device) if return_dist else None
    _knn_xyz(b, n, m, int(K), pts, queries, idx, dist, int(arith))
    if return_dist:
        return idx.to(torch.int64), dist
    return idx.to(torch.int64)


def knn_query(k, support_pts, query_pts):
    """Common/ops.py:110-118: neighbour indices [B,N2,k] as int32."""
    pts, queries = f32(support_pts, "support_pts"), f32(query_pts, "query_pts")
    b, n, _ = pts.shape
    m = queries.shape[1]
    req(0 < int(k) <= n and int(k) <= 4096, "knn_query supports 1 <= k <= min(N1, 4096)")
    idx = torch.empty((b, m, int(k)), dtype=torch.int32, device=pts.device)
    _knn_xyz(b, n, m, int(k), pts, queries, idx, None, _lib.ARITH_PLAIN)
    return idx

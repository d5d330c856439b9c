"""Ball query / grouping / k-NN helpers -- the reference wrapper tf_ops/grouping/tf_grouping.py:9-141
with torch tensors.  Names, positional order and return order are the reference's."""
import numbers

import torch

from . import _lib
from ._util import f32, i32, req


def query_ball_point(radius, nsample, xyz, new_xyz, arith=_lib.ARITH_CONTRACT):
    """(radius: float | [b] f32, nsample, xyz[b,n,3], new_xyz[b,m,3]) -> (idx[b,m,ns] i32, pts_cnt[b,m] i32)
    tf_grouping.py:9-30.  Only radius[0] is used (tf_grouping_g.cu:25); the first nsample dataset
    indices inside the ball in INDEX order, padded with the first hit.  Rows whose query has no
    neighbour are returned as zeros (the reference leaves them uninitialised)."""
    xyz, new_xyz = f32(xyz, "xyz"), f32(new_xyz, "new_xyz")
    req(xyz.dim() == 3 and xyz.shape[2] == 3, "QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
    req(new_xyz.dim() == 3 and new_xyz.shape[2] == 3 and new_xyz.shape[0] == xyz.shape[0],
        "QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
    req(int(nsample) > 0, "QueryBallPoint expects positive nsample")
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    if isinstance(radius, numbers.Number):
        req(radius > 0, "QueryBallPoint expects positive radius")
        radius = torch.full((max(b, 1),), float(radius), dtype=torch.float32, device=xyz.device)
    else:
        radius = f32(radius, "radius").reshape(-1)
    idx = torch.zeros((b, m, int(nsample)), dtype=torch.int32, device=xyz.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz.device)
    _lib.check(_lib.lib().dispu_query_ball(b, n, m, _lib.ptr(radius), int(nsample), _lib.ptr(xyz), _lib.ptr(new_xyz),
                                           _lib.ptr(idx), _lib.ptr(cnt), int(arith), _lib.stream_ptr(xyz.device)),
               "dispu_query_ball")
    return idx, cnt


def select_top_k(k, dist):
    """(k, dist[b,m,n] f32) -> (idx[b,m,n] i32, dist_out[b,m,n] f32): the first k entries of every row are the k SMALLEST
    distances in ascending order with their positions, the rest is what k rounds of selection sort leave behind
    (tf_grouping.py:33-42 -> SelectionSort, tf_grouping.cpp:112-143; optional op, unused by the shipped graph; no gradient)."""
    dist = f32(dist, "dist")
    req(int(k) > 0, "SelectionSort expects positive k")
    req(dist.dim() == 3, "SelectionSort expects (b,m,n) dist shape.")
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    if b * m * n:
        _lib.check(_lib.lib().dispu_selection_sort(b, n, m, int(k), _lib.ptr(dist), _lib.ptr(outi), _lib.ptr(out),
                                                   _lib.stream_ptr(dist.device)), "dispu_selection_sort")
    return outi, out


class _GroupPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
        _lib.check(_lib.lib().dispu_group_point(b, n, c, m, ns, _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out),
                                                _lib.stream_ptr(points.device)), "dispu_group_point")
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _group_point_grad_raw(ctx.n, idx, grad_out.contiguous()), None


def _group_point_grad_raw(n, idx, grad_out):
    b, m, ns, c = grad_out.shape
    g = torch.empty((b, n, c), dtype=torch.float32, device=grad_out.device)
    _lib.check(_lib.lib().dispu_group_point_grad(b, n, c, m, ns, _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(g),
                                                 _lib.stream_ptr(grad_out.device)), "dispu_group_point_grad")
    return g


def group_point(points, idx):
    """(points[b,n,c], idx[b,m,ns] i32) -> [b,m,ns,c].   tf_grouping.py:44-52; gradient :53-57."""
    points, idx = f32(points, "points"), i32(idx, "idx")
    req(points.dim() == 3, "GroupPoint expects (batch_size, num_points, channel) points shape")
    req(idx.dim() == 3 and idx.shape[0] == points.shape[0], "GroupPoint expects (batch_size, npoints, nsample) idx shape")
    return _GroupPoint.apply(points, idx)


def group_point_grad(points, idx, grad_out):
    """grouping_module.group_point_grad(points, idx, grad_out) -> [b,n,c]   (tf_grouping.py:53-57)."""
    points, idx, grad_out = f32(points, "points"), i32(idx, "idx"), f32(grad_out, "grad_out")
    req(points.dim() == 3, "GroupPointGrad expects (batch_size, num_points, channel) points shape")
    req(idx.dim() == 3 and idx.shape[0] == points.shape[0], "GroupPointGrad expects (batch_size, npoints, nsample) idx shape")
    req(grad_out.dim() == 4 and tuple(grad_out.shape[:3]) == tuple(idx.shape) and grad_out.shape[3] == points.shape[2],
        "GroupPointGrad expects (batch_size, npoints, nsample, channel) grad_out shape")
    return _group_point_grad_raw(points.shape[1], idx, grad_out)


def knn_point(k, xyz1, xyz2):
    """(k, xyz1[b,n,c] dataset, xyz2[b,m,c] queries) -> (val[b,m,k] = NEGATIVE squared distance, idx[b,m,k] i32)
    tf_grouping.py:116-141 (top_k(-dist): ascending distance, ties -> lower index)."""
    xyz1, xyz2 = f32(xyz1, "xyz1"), f32(xyz2, "xyz2")
    req(xyz1.dim() == 3 and xyz2.dim() == 3 and xyz1.shape[0] == xyz2.shape[0] and xyz1.shape[2] == xyz2.shape[2],
        "knn_point expects xyz1 (b,n,c) and xyz2 (b,m,c)")
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    req(0 < int(k) <= n, "input must have at least k columns")  # tf.nn.top_k's own error text
    val = torch.empty((b, m, int(k)), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, m, int(k)), dtype=torch.int32, device=xyz1.device)
    _lib.check(_lib.lib().dispu_knn_point(b, n, m, c, int(k), _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(val), _lib.ptr(idx),
                                          _lib.stream_ptr(xyz1.device)), "dispu_knn_point")
    return val, idx


def knn_point_2(k, points, queries, sort=True, unique=True):
    """(k, points[b,P0,C], queries[b,P,C]) -> (dist[b,P,k], indices[b,P,k,2] i32 = (batch, point))
    tf_grouping.py:95-114.  `unique` is accepted and ignored: it is a no-op in the reference (:89-91);
    `sort=False` also returns the sorted order (a valid top_k answer)."""
    points, queries = f32(points, "points"), f32(queries, "queries")
    req(points.dim() == 3 and queries.dim() == 3 and points.shape[0] == queries.shape[0]
        and points.shape[2] == queries.shape[2], "knn_point_2 expects points (N,P0,C) and queries (N,P,C)")
    b, n, c = points.shape
    m = queries.shape[1]
    req(0 < int(k) <= n, "input must have at least k columns")
    dist = torch.empty((b, m, int(k)), dtype=torch.float32, device=points.device)
    idx = torch.empty((b, m, int(k)), dtype=torch.int32, device=points.device)
    L = _lib.lib()
    nbytes = L.dispu_knn_feat_scratch_bytes(b, n, m, c, int(k))          # clouds of 513 .. 4096 points: chunked wave search
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=points.device) if nbytes else None
    _lib.check(L.dispu_knn_feat_strided_ws(b, n, m, c, int(k), _lib.ptr(points), c, _lib.ptr(queries), c, _lib.ptr(dist), _lib.ptr(idx),
                                           _lib.ptr(scratch), nbytes, _lib.stream_ptr(points.device)), "dispu_knn_feat")
    bidx = torch.arange(b, dtype=torch.int32, device=points.device).view(b, 1, 1).expand(b, m, int(k))
    return dist, torch.stack([bidx, idx], dim=3)

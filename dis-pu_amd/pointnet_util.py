"""PointNet++ set-abstraction / feature-propagation modules -- the reference's Common/pointnet_util.py:22-222 with
torch tensors, composed from the hot-path ops (FPS, gather, ball query / k-NN, group, 3-NN, interpolate) and the
fused 1x1-conv GEMM.  Same function names, argument order and return values; variables are passed through
`params` (see tf_util.py) instead of TF variable scopes.  is_training=True runs BatchNorm on batch statistics and updates
the moving statistics in `params` (forward only: the hand-written backward exists for the generator, train.py).

torch is used here only to allocate / concatenate / reshape device buffers; every arithmetic step is a HIP kernel
of libdispu_hip.so."""
import ctypes
import os

import torch

from . import _lib, tf_util
from .tf_grouping import group_point, knn_point, query_ball_point
from .tf_interpolate import three_interpolate, three_nn
from .tf_sampling import farthest_point_sample, gather_point

FUSED_SA = True               # set abstraction as one launch (dispu_sa_fused); False: the single ops (A/B tests)

_POOL = {"max": 0, "avg": 1, "min": 2, "weighted_avg": 3, "max_and_avg": 4}


def _center(grouped, center):
    """grouped[b,m,ns,c] -= center[b,m,c] in place (pointnet_util.py:43)."""
    b, m, ns, c = grouped.shape
    _lib.check(_lib.lib().dispu_group_center(b * m, ns, c, _lib.ptr(grouped), _lib.ptr(center.contiguous()),
                                             _lib.stream_ptr(grouped.device)), "dispu_group_center")
    return grouped


def _pool(x, pooling, grouped_xyz=None):
    b, m, ns, c = x.shape
    mode = _POOL[pooling]
    out = torch.empty((b, m, 1, 2 * c if mode == 4 else c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dispu_pool_nsample(b * m, ns, c, mode, _lib.ptr(x.contiguous()),
                                             _lib.ptr(grouped_xyz.contiguous()) if mode == 3 else None, _lib.ptr(out),
                                             _lib.stream_ptr(x.device)), "dispu_pool_nsample")
    return out


def sample_and_group(npoint, radius, nsample, xyz, points, tnet_spec=None, knn=False, use_xyz=True):
    """pointnet_util.py:22-59 -> (new_xyz[b,npoint,3], new_points[b,npoint,ns,3+c], idx[b,npoint,ns], grouped_xyz)."""
    if tnet_spec is not None:
        raise NotImplementedError("tnet is undefined in the reference as well (pointnet_util.py:45 calls a missing symbol)")
    new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
    if knn:
        _, idx = knn_point(nsample, xyz, new_xyz)
    else:
        idx, _ = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = _center(group_point(xyz, idx), new_xyz)
    if points is not None:
        grouped_points = group_point(points, idx)
        new_points = torch.cat([grouped_xyz, grouped_points], dim=-1) if use_xyz else grouped_points
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def sample_and_group_all(xyz, points, use_xyz=True):
    """pointnet_util.py:62-88: one group holding every point, centroid (0,0,0)."""
    b, n, _ = xyz.shape
    new_xyz = torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device)
    idx = torch.arange(n, dtype=torch.int32, device=xyz.device).view(1, 1, n).expand(b, 1, n).contiguous()
    grouped_xyz = xyz.reshape(b, 1, n, 3)
    if points is not None:
        new_points = (torch.cat([xyz, points], dim=2) if use_xyz else points).unsqueeze(1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def _sa_fused(xyz, new_xyz, points, idx, mlp, scope, params, bn):
    """group -> centre -> MLP -> max over nsample in ONE launch (csrc/sa_fused.hip): [b, m, ns, C] never reaches HBM."""
    b, n, _ = xyz.shape
    m, ns = idx.shape[1], idx.shape[2]
    dev = xyz.device
    c = 0 if points is None else points.shape[2]
    keep, Ws, bs, scs, shs = [], [], [], [], []
    cin = 3 + c
    for i, co in enumerate(mlp):
        sc = scope + "/conv%d" % i
        W, bias = tf_util._dev(params[sc + "/weights"], dev), tf_util._dev(params[sc + "/biases"], dev)
        if tuple(W.shape) != (cin, co):
            raise ValueError("%s/weights has shape %s, expected (%d, %d)" % (sc, tuple(W.shape), cin, co))
        scale, shift = tf_util.bn_fold(params, sc, dev) if bn else (None, None)
        keep += [W, bias, scale, shift]
        Ws.append(_lib.ptr(W)); bs.append(_lib.ptr(bias)); scs.append(_lib.ptr(scale)); shs.append(_lib.ptr(shift))
        cin = co
    arr = lambda ps: (ctypes.c_void_p * len(ps))(*[p.value for p in ps])       # NULL entries: no BatchNorm on that layer
    couts = (ctypes.c_int * len(mlp))(*[int(v) for v in mlp])
    out = torch.empty((b, m, mlp[-1]), dtype=torch.float32, device=dev)
    pts = points.contiguous() if points is not None else None
    _lib.check(_lib.lib().dispu_sa_fused(b, n, m, ns, c, _lib.ptr(xyz.contiguous()), _lib.ptr(new_xyz.contiguous()), _lib.ptr(pts),
                                         _lib.ptr(idx.contiguous()), len(mlp), arr(Ws), arr(bs), arr(scs), arr(shs), couts, _lib.ptr(out),
                                         _lib.stream_ptr(dev)), "dispu_sa_fused")
    return out


def _sa_fusable(points, nsample, mlp, mlp2, group_all, is_training, bn, pooling, tnet_spec, use_xyz):
    if not FUSED_SA or group_all or mlp2 or pooling != "max" or tnet_spec is not None or not use_xyz:
        return False
    if (bn and is_training) or nsample not in (32, 64) or not 1 <= len(mlp) <= 3:
        return False
    c = 0 if points is None else points.shape[2]
    width = max([(3 + c + 1) & ~1] + [(co + 1) & ~1 for co in mlp[:-1]]) | 1
    return (2 * nsample * width + (nsample // 16) * mlp[-1]) * 4 <= 160 * 1024


def pointnet_sa_module(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, is_training, bn_decay, scope,
                       bn=True, pooling="max", tnet_spec=None, knn=False, use_xyz=True, params=None):
    """pointnet_util.py:91-149 -> (new_xyz, new_points[b,npoint,mlp[-1] or mlp2[-1]], idx).

    Max pooling over 32 / 64 samples without mlp2 in inference mode (every call of Common/ops.py:505-550) runs as ONE fused
    kernel after the sampling / ball query; other configurations compose the single ops (pointnet_util.FUSED_SA = False forces that)."""
    if _sa_fusable(points, nsample, mlp, mlp2, group_all, is_training, bn, pooling, tnet_spec, use_xyz):
        new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
        if knn:
            _, idx = knn_point(nsample, xyz, new_xyz)
        else:
            idx, _ = query_ball_point(radius, nsample, xyz, new_xyz)
        return new_xyz, _sa_fused(xyz, new_xyz, points, idx, mlp, scope, params, bn), idx
    if group_all:
        nsample = xyz.shape[1]
        new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(xyz, points, use_xyz)
    else:
        new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, nsample, xyz, points, tnet_spec, knn, use_xyz)
    for i, co in enumerate(mlp):
        new_points = tf_util.conv2d(new_points, co, (1, 1), scope + "/conv%d" % i, params, bn=bn, is_training=is_training, bn_decay=bn_decay)
    new_points = _pool(new_points, pooling, grouped_xyz)
    for i, co in enumerate(mlp2 or []):
        new_points = tf_util.conv2d(new_points, co, (1, 1), scope + "/conv_post_%d" % i, params, bn=bn, is_training=is_training, bn_decay=bn_decay)
    return new_xyz, new_points.squeeze(2), idx


def pointnet_sa_module_msg(xyz, points, npoint, radius_list, nsample_list, mlp_list, is_training=False, bn_decay=None,
                           scope="msg", bn=True, use_xyz=True, params=None):
    """pointnet_util.py:152-189 (multi-scale grouping; note the [points, xyz] concat order of this variant)."""
    new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
    outs = []
    for i, (radius, nsample) in enumerate(zip(radius_list, nsample_list)):
        idx, _ = query_ball_point(radius, nsample, xyz, new_xyz)
        grouped_xyz = _center(group_point(xyz, idx), new_xyz)
        if points is not None:
            grouped_points = group_point(points, idx)
            if use_xyz:
                grouped_points = torch.cat([grouped_points, grouped_xyz], dim=-1)
        else:
            grouped_points = grouped_xyz
        for j, co in enumerate(mlp_list[i]):
            grouped_points = tf_util.conv2d(grouped_points, co, (1, 1), scope + "/conv%d_%d" % (i, j), params, bn=bn,
                                            is_training=is_training, bn_decay=bn_decay)
        outs.append(_pool(grouped_points, "max").squeeze(2))
    return new_xyz, torch.cat(outs, dim=-1)


def pointnet_fp_module(xyz1, xyz2, points1, points2, mlp, is_training, bn_decay, scope, bn=True, params=None):
    """pointnet_util.py:192-222: 3-NN inverse-distance interpolation of points2 onto xyz1, concat, MLP."""
    dist, idx = three_nn(xyz1, xyz2)
    weight = torch.empty_like(dist)
    _lib.check(_lib.lib().dispu_idw_weights(dist.shape[0] * dist.shape[1], _lib.ptr(dist), _lib.ptr(weight),
                                            _lib.stream_ptr(dist.device)), "dispu_idw_weights")
    interpolated = three_interpolate(points2, idx, weight)
    new_points1 = torch.cat([interpolated, points1], dim=2) if points1 is not None else interpolated
    new_points1 = new_points1.unsqueeze(2)
    for i, co in enumerate(mlp):
        new_points1 = tf_util.conv2d(new_points1, co, (1, 1), scope + "/conv_%d" % i, params, bn=bn, is_training=is_training, bn_decay=bn_decay)
    return new_points1.squeeze(2)

"""DeepGCN vertex / edge layers that the reference wires up (gcn_lib/tf_vertex.py:81-101 `edge_conv_layer`,
gcn_lib/tf_edge.py:19-28 `knn_graph`, Common/tf_util.py:618-686 pairwise_distance / knn / get_edge_feature),
gcn_lib/tf_vertex.py:20-79 `max_relat_conv_layer`, :103-180 `graphsage_conv_layer`, :182-251 `gin_conv_layer`),
on the hot-path kernels.  `nn.build` of the reference (gcn_lib/tf_nn.py:37-56) is a single tf_util.conv2d; its
options are keyword arguments here.  Dilated graphs are never instantiated by the reference and are not built."""
import ctypes
import os

import torch

from . import _lib, tf_util
from .tf_grouping import group_point, knn_point_2

FUSED_EDGE_CONV = True       # EdgeConv as one launch (dispu_edge_conv_fused); False: the unfused composition (A/B tests)


def knn_graph(vertex_features, k):
    """tf_edge.py:19-28: neighbour indices [b, n, k] from `||x||^2 - 2 x.x^T + ||x||^2` + top_k(-D, k)
    (tf_util.py:618-651; the point itself is NOT dropped here).  vertex_features [b, n, c] or [b, n, 1, c]."""
    f = vertex_features.squeeze(2) if vertex_features.dim() == 4 else vertex_features
    _, idx = knn_point_2(k, f, f)
    return idx[..., 1].contiguous()


def get_edge_feature(point_cloud, nn_idx, k):
    """tf_util.py:654-686: [b, n, k, 2c] = [central | neighbour - central]."""
    f = (point_cloud.squeeze(2) if point_cloud.dim() == 4 else point_cloud).contiguous()
    b, n, c = f.shape
    out = torch.empty((b, n, k, 2 * c), dtype=torch.float32, device=f.device)
    idx = nn_idx.contiguous()
    _lib.check(_lib.lib().dispu_edge_feature(b * n, n, k, c, _lib.ptr(f), c, _lib.ptr(idx), idx.shape[-1], 0, _lib.ptr(out),
                                             2 * c, _lib.stream_ptr(f.device)), "dispu_edge_feature")
    return out


def edge_conv_layer(inputs, neigh_idx, k, num_outputs, scope=None, is_training=False, params=None, bn=False,
                    activation_fn="relu"):
    """tf_vertex.py:81-101 (EdgeConv): MLP over [x_i, x_j - x_i], max over the k neighbours, keep_dims.
    `nn.build` of the reference (gcn_lib/tf_nn.py:37-56) is a single tf_util.conv2d; its options are the keyword
    arguments here.  Returns [b, n, 1, num_outputs]."""
    f = _squeeze(inputs)
    c = f.shape[-1]
    width = ((2 * c + 1) & ~1) | 1
    if (k in (16, 32, 64) and not (bn and is_training) and FUSED_EDGE_CONV
            and (2 * 64 * width + 4 * num_outputs) * 4 <= 160 * 1024):
        # edge feature, the conv and the max over the neighbours in ONE launch (csrc/sa_fused.hip): no [b, n, k, 2c] tensor in HBM
        b, n, _ = f.shape
        dev = f.device
        W, bias = tf_util._dev(params[scope + "/weights"], dev), tf_util._dev(params[scope + "/biases"], dev)
        if tuple(W.shape) != (2 * c, num_outputs):
            raise ValueError("%s/weights has shape %s, expected (%d, %d)" % (scope, tuple(W.shape), 2 * c, num_outputs))
        scale, shift = tf_util.bn_fold(params, scope, dev) if bn else (None, None)
        idx = neigh_idx.contiguous()
        one = lambda t: (ctypes.c_void_p * 1)(_lib.ptr(t).value)
        pooled = torch.empty((b, n, 1, num_outputs), dtype=torch.float32, device=dev)
        act = {"relu": 1, None: 0, "none": 0}[activation_fn]
        _lib.check(_lib.lib().dispu_edge_conv_fused(b, n, k, c, _lib.ptr(f), c, _lib.ptr(idx), idx.shape[-1], 1, one(W), one(bias), one(scale),
                                                    one(shift), (ctypes.c_int * 1)(int(num_outputs)), act, _lib.ptr(pooled),
                                                    _lib.stream_ptr(dev)), "dispu_edge_conv_fused")
        return pooled
    edge = get_edge_feature(inputs, neigh_idx, k)
    out = tf_util.conv2d(edge, num_outputs, (1, 1), scope, params, bn=bn, is_training=is_training, activation_fn=activation_fn)
    b, n, _, co = out.shape
    pooled = torch.empty((b, n, 1, co), dtype=torch.float32, device=out.device)
    _lib.check(_lib.lib().dispu_pool_nsample(b * n, k, co, 0, _lib.ptr(out), None, _lib.ptr(pooled),
                                             _lib.stream_ptr(out.device)), "dispu_pool_nsample")
    return pooled


def _squeeze(inputs):
    return (inputs.squeeze(2) if inputs.dim() == 4 else inputs).contiguous()


def _pool(x, k, mode):
    """x [b, n, k, c] -> [b, n, 1, c]: 0 max, 5 sum over the neighbour axis (dispu_pool_nsample)."""
    b, n, _, c = x.shape
    out = torch.empty((b, n, 1, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dispu_pool_nsample(b * n, k, c, mode, _lib.ptr(x.contiguous()), None, _lib.ptr(out),
                                             _lib.stream_ptr(x.device)), "dispu_pool_nsample")
    return out


def get_max_relat_feature(inputs, neigh_idx, k):
    """tf_vertex.py:40-79 (MRGCN): [b, n, 1, 2c] = [x_i | max_j (x_j - x_i)].  The edge feature kernel already forms
    [x_i | x_j - x_i] per pair; the max over the k copies of x_i is x_i itself."""
    return _pool(get_edge_feature(inputs, neigh_idx, k), k, 0)


def max_relat_conv_layer(inputs, neigh_idx, k, num_outputs, scope=None, is_training=False, params=None, bn=False,
                         activation_fn="relu"):
    """tf_vertex.py:20-38.  Returns [b, n, 1, num_outputs]."""
    return tf_util.conv2d(get_max_relat_feature(inputs, neigh_idx, k), num_outputs, (1, 1), scope, params, bn=bn,
                          is_training=is_training, activation_fn=activation_fn)


def get_graphsage_feature(inputs, neigh_idx, k, scope=None, is_training=False, params=None, bn=False, activation_fn="relu"):
    """tf_vertex.py:136-180: [b, n, 1, 2c] = [x_i | max_j MLP(x_j)] with an MLP c -> c on the gathered neighbours."""
    f = _squeeze(inputs)
    c = f.shape[-1]
    nbr = group_point(f, neigh_idx[..., :k].contiguous())                      # [b, n, k, c]
    h = tf_util.conv2d(nbr, c, (1, 1), scope, params, bn=bn, is_training=is_training, activation_fn=activation_fn)
    return torch.cat([f.unsqueeze(2), _pool(h, k, 0)], dim=-1)


def graphsage_conv_layer(inputs, neigh_idx, k, num_outputs, normalize=True, scope=None, is_training=False, params=None,
                         bn=False, activation_fn="relu"):
    """tf_vertex.py:103-134: MLP over the GraphSAGE feature (aggregator weights under scope + '_aggr'), then
    tf.nn.l2_normalize over the channel axis."""
    aggr = get_graphsage_feature(inputs, neigh_idx, k, scope=scope + "_aggr", is_training=is_training, params=params, bn=bn,
                                 activation_fn=activation_fn)
    out = tf_util.conv2d(aggr, num_outputs, (1, 1), scope, params, bn=bn, is_training=is_training, activation_fn=activation_fn)
    if not normalize:
        return out
    b, n, _, co = out.shape
    res = torch.empty_like(out)
    _lib.check(_lib.lib().dispu_l2_normalize_rows(b * n, co, _lib.ptr(out.contiguous()), _lib.ptr(res), _lib.stream_ptr(out.device)),
               "dispu_l2_normalize_rows")
    return res


def get_gin_feature(inputs, neigh_idx, k):
    """tf_vertex.py:218-251: [b, n, 1, c] = sum_j x_j over the k neighbours."""
    f = _squeeze(inputs)
    return _pool(group_point(f, neigh_idx[..., :k].contiguous()), k, 5)


def gin_conv_layer(inputs, neigh_idx, k, num_outputs, zero_epsilon=False, scope=None, is_training=False, params=None, bn=False,
                   activation_fn="relu"):
    """tf_vertex.py:182-216: MLP over x_i (1 + epsilon) + sum_j x_j; epsilon = params[scope + '_epsilon'] (a scalar variable,
    zero-initialised in the reference; `zero_epsilon` only decides whether it is trainable there)."""
    f = _squeeze(inputs).unsqueeze(2).contiguous()
    aggr = get_gin_feature(inputs, neigh_idx, k)
    eps = 0.0
    if params is not None and (scope + "_epsilon") in params:
        e = params[scope + "_epsilon"]
        eps = float(e.reshape(-1)[0]) if hasattr(e, "reshape") else float(e)
    comb = torch.empty_like(aggr)
    _lib.check(_lib.lib().dispu_scale_add(comb.numel(), _lib.ptr(f), 1.0 + eps, _lib.ptr(aggr), _lib.ptr(comb),
                                          _lib.stream_ptr(f.device)), "dispu_scale_add")
    return tf_util.conv2d(comb, num_outputs, (1, 1), scope, params, bn=bn, is_training=is_training, activation_fn=activation_fn)

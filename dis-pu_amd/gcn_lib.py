"""DeepGCN vertex / edge layers that the reference wires up (gcn_lib/tf_vertex.py:81-101 `edge_conv_layer`,
gcn_lib/tf_edge.py:19-28 `knn_graph`, Common/tf_util.py:618-686 pairwise_distance / knn / get_edge_feature),
on the hot-path kernels.  MRGCN / GraphSAGE / GIN / dilated graphs are never instantiated by the reference
 and are not built."""
import torch

from . import _lib, tf_util
from .tf_grouping import knn_point_2


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
    edge = get_edge_feature(inputs, neigh_idx, k)
    out = tf_util.conv2d(edge, num_outputs, (1, 1), scope, params, bn=bn, is_training=is_training, activation_fn=activation_fn)
    b, n, _, co = out.shape
    pooled = torch.empty((b, n, 1, co), dtype=torch.float32, device=out.device)
    _lib.check(_lib.lib().dispu_pool_nsample(b * n, k, co, 0, _lib.ptr(out), None, _lib.ptr(pooled),
                                             _lib.stream_ptr(out.device)), "dispu_pool_nsample")
    return pooled

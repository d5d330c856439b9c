"""CPU ORACLE (test infrastructure) for the compositions built on the hot-path ops: PointNet++ modules
(Common/pointnet_util.py:22-222), EdgeConv / kNN graph (gcn_lib/tf_vertex.py:81-101, tf_edge.py:19-28,
Common/tf_util.py:618-686) and the training losses (Common/loss_utils.py:45-84,170-176,271-298).
numpy restatements over oracle/oracle.py and oracle/generator.py:linear; imported by tests only."""
import numpy as np

from . import generator as OG
from . import oracle as O

BN_EPS = 1e-3


def conv2d(x, P, scope, bn=False, relu=True):
    """tf_util.conv2d 1x1 (tf_util.py:120-185): conv -> bias_add -> [batch_norm, inference] -> [relu]."""
    y = OG.linear(x, P[scope + "/weights"], P[scope + "/biases"], relu=False)
    if bn:
        g, b = P[scope + "/bn/gamma"].astype(np.float64), P[scope + "/bn/beta"].astype(np.float64)
        mu, var = P[scope + "/bn/moving_mean"].astype(np.float64), P[scope + "/bn/moving_variance"].astype(np.float64)
        scale = (g / np.sqrt(var + BN_EPS)).astype(np.float32)
        shift = (b - mu * (g / np.sqrt(var + BN_EPS))).astype(np.float32)
        y = y * scale + shift
    return np.maximum(y, np.float32(0)) if relu else y


def sample_and_group(npoint, radius, nsample, xyz, points, knn=False, use_xyz=True):
    """pointnet_util.py:22-59"""
    new_xyz = O.gather_point(xyz, O.farthest_point_sample(npoint, xyz))
    if knn:
        _, idx = O.knn_point(nsample, xyz, new_xyz)
    else:
        idx, _ = O.query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = O.group_point(xyz, idx) - new_xyz[:, :, None, :]
    if points is not None:
        gp = O.group_point(points, idx)
        new_points = np.concatenate([grouped_xyz, gp], -1) if use_xyz else gp
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def pool(x, pooling, grouped_xyz=None):
    """pointnet_util.py:121-140 (keep_dims)."""
    if pooling == "max":
        return x.max(2, keepdims=True)
    if pooling == "avg":
        return (np.add.reduce(x.astype(np.float64), 2, keepdims=True) / x.shape[2]).astype(np.float32)
    if pooling == "min":
        return (-x).max(2, keepdims=True)          # the reference returns max(-x), it never negates back
    if pooling == "weighted_avg":
        d = np.sqrt((grouped_xyz.astype(np.float64) ** 2).sum(-1, keepdims=True))
        e = np.exp(-d * 5)
        return (x * (e / e.sum(2, keepdims=True))).sum(2, keepdims=True).astype(np.float32)
    if pooling == "max_and_avg":
        return np.concatenate([x.max(2, keepdims=True),
                               (np.add.reduce(x.astype(np.float64), 2, keepdims=True) / x.shape[2]).astype(np.float32)], -1)
    raise ValueError(pooling)


def pointnet_sa_module(P, scope, xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, bn=True, pooling="max",
                       knn=False, use_xyz=True):
    """pointnet_util.py:91-149"""
    if group_all:
        b, n, _ = xyz.shape
        new_xyz = np.zeros((b, 1, 3), np.float32)
        idx = np.broadcast_to(np.arange(n, dtype=np.int32)[None, None], (b, 1, n))
        grouped_xyz = xyz.reshape(b, 1, n, 3)
        new_points = (np.concatenate([xyz, points], 2) if use_xyz else points)[:, None] if points is not None else grouped_xyz
    else:
        new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, nsample, xyz, points, knn, use_xyz)
    for i in range(len(mlp)):
        new_points = conv2d(new_points, P, scope + "/conv%d" % i, bn=bn)
    new_points = pool(new_points, pooling, grouped_xyz)
    for i in range(len(mlp2 or [])):
        new_points = conv2d(new_points, P, scope + "/conv_post_%d" % i, bn=bn)
    return new_xyz, new_points[:, :, 0], idx


def pointnet_sa_module_msg(P, scope, xyz, points, npoint, radius_list, nsample_list, mlp_list, bn=True, use_xyz=True):
    """pointnet_util.py:152-189"""
    new_xyz = O.gather_point(xyz, O.farthest_point_sample(npoint, xyz))
    outs = []
    for i, (r, ns) in enumerate(zip(radius_list, nsample_list)):
        idx, _ = O.query_ball_point(r, ns, xyz, new_xyz)
        gxyz = O.group_point(xyz, idx) - new_xyz[:, :, None, :]
        if points is not None:
            gp = O.group_point(points, idx)
            if use_xyz:
                gp = np.concatenate([gp, gxyz], -1)
        else:
            gp = gxyz
        for j in range(len(mlp_list[i])):
            gp = conv2d(gp, P, scope + "/conv%d_%d" % (i, j), bn=bn)
        outs.append(gp.max(2))
    return new_xyz, np.concatenate(outs, -1)


def pointnet_fp_module(P, scope, xyz1, xyz2, points1, points2, mlp, bn=True):
    """pointnet_util.py:192-222"""
    dist, idx = O.three_nn(xyz1, xyz2)
    inv = np.float32(1.0) / np.maximum(dist, np.float32(1e-10))
    norm = (inv[..., 0:1] + inv[..., 1:2]) + inv[..., 2:3]
    weight = inv / norm
    interp = O.three_interpolate(points2, idx, weight)
    x = np.concatenate([interp, points1], 2) if points1 is not None else interp
    x = x[:, :, None]
    for i in range(len(mlp)):
        x = conv2d(x, P, scope + "/conv_%d" % i, bn=bn)
    return x[:, :, 0]


def hierachy_feature_extractor(P, inputs, npoints=(1024, 384, 128), radius=(0.1, 0.2, 0.4)):
    """Common/ops.py:505-550 (modules with their default bn=True; scopes layer1..4, fa_layer1..4) -> [B, N, 128]; `tap`-free."""
    l0_xyz, l0_points = inputs, None
    l1_xyz, l1_points, _ = pointnet_sa_module(P, "layer1", l0_xyz, l0_points, npoints[0], radius[0], 64, [32, 32, 64], None, False)
    l2_xyz, l2_points, _ = pointnet_sa_module(P, "layer2", l1_xyz, l1_points, npoints[1], radius[1], 64, [64, 64, 128], None, False)
    l3_xyz, l3_points, _ = pointnet_sa_module(P, "layer3", l2_xyz, l2_points, npoints[2], radius[2], 64, [128, 128, 256], None, False)
    l4_xyz, l4_points, _ = pointnet_sa_module(P, "layer4", l3_xyz, l3_points, None, None, None, [256, 256, 512], None, True)
    l3_points = pointnet_fp_module(P, "fa_layer1", l3_xyz, l4_xyz, l3_points, l4_points, [512, 512])
    l2_points = pointnet_fp_module(P, "fa_layer2", l2_xyz, l3_xyz, l2_points, l3_points, [512, 256])
    l1_points = pointnet_fp_module(P, "fa_layer3", l1_xyz, l2_xyz, l1_points, l2_points, [256, 128])
    return pointnet_fp_module(P, "fa_layer4", l0_xyz, l1_xyz, l0_points, l1_points, [128, 128, 128])


def knn_graph(feat, k):
    """tf_edge.py:19-28 + tf_util.py:618-651 (self is kept)."""
    return O.knn_point_2(k, feat, feat)[1][..., 1]


def edge_conv_layer(P, scope, feat, idx, bn=False, relu=True):
    """tf_vertex.py:81-101"""
    nbr = OG.gather(feat, idx)
    central = np.broadcast_to(feat[:, :, None, :], nbr.shape)
    edge = np.concatenate([central, nbr - central], -1)
    return conv2d(edge, P, scope, bn=bn, relu=relu).max(2, keepdims=True)


def max_relat_conv_layer(P, scope, feat, idx, bn=False, relu=True):
    """tf_vertex.py:20-79 (MRGCN): conv over [x_i | max_j (x_j - x_i)]."""
    nbr = OG.gather(feat, idx)
    rel = (nbr - feat[:, :, None, :]).max(2, keepdims=True)
    return conv2d(np.concatenate([feat[:, :, None, :], rel], -1), P, scope, bn=bn, relu=relu)


def graphsage_conv_layer(P, scope, feat, idx, normalize=True, bn=False, relu=True):
    """tf_vertex.py:103-180: conv over [x_i | max_j conv_aggr(x_j)], then l2_normalize over channels
    (x * rsqrt(max(sum x^2, 1e-12)), the sum in channel order)."""
    h = conv2d(OG.gather(feat, idx), P, scope + "_aggr", bn=bn, relu=relu).max(2, keepdims=True)
    out = conv2d(np.concatenate([feat[:, :, None, :], h], -1), P, scope, bn=bn, relu=relu)
    if not normalize:
        return out
    ss = np.zeros(out.shape[:-1], np.float32)
    for c in range(out.shape[-1]):
        ss = ss + out[..., c] * out[..., c]
    inv = np.float32(1.0) / np.sqrt(np.maximum(ss, np.float32(1e-12)))
    return out * inv[..., None]


def gin_conv_layer(P, scope, feat, idx, bn=False, relu=True):
    """tf_vertex.py:182-251: conv over x_i (1 + epsilon) + sum_j x_j (the sum in neighbour order)."""
    nbr = OG.gather(feat, idx)
    agg = np.zeros(feat.shape, np.float32)
    for s in range(nbr.shape[2]):
        agg = agg + nbr[:, :, s, :]
    eps = np.float32(np.asarray(P.get(scope + "_epsilon", 0.0)).reshape(-1)[0])
    comb = feat * (np.float32(1.0) + eps) + agg
    return conv2d(comb[:, :, None, :], P, scope, bn=bn, relu=relu)


def chamfer(pred, gt, radius=1.0, forward_weight=1.0):
    """loss_utils.py:45-64"""
    d1, _, d2, _ = O.nn_distance(gt, pred)
    cd = forward_weight * d1.astype(np.float64).mean(1) + d2.astype(np.float64).mean(1)
    return float((cd / radius).mean())


def hausdorff_loss(pred, gt, radius=1.0, forward_weight=1.0):
    """loss_utils.py:67-84"""
    d1, _, d2, _ = O.nn_distance(gt, pred)
    return float(((forward_weight * d1.max(1) + d2.max(1)) / radius).max())


def earth_mover(pcd1, pcd2, radius=1.0):
    """loss_utils.py:170-176"""
    m = O.approx_match(pcd1, pcd2)
    cost = O.match_cost(pcd1, pcd2, m).astype(np.float64) / radius
    return float((cost / pcd1.shape[1]).mean())


def get_repulsion_loss(pred, nsample=20, radius=0.07, h=0.001):
    """loss_utils.py:271-298 (knn=False, use_l1=False)."""
    idx, _ = O.query_ball_point(radius, nsample, pred, pred)
    g = O.group_point(pred, idx) - pred[:, :, None, :]
    d = ((g[..., 0] * g[..., 0] + g[..., 1] * g[..., 1]) + g[..., 2] * g[..., 2]).astype(np.float32)
    order = np.argsort(d, axis=-1, kind="stable")[..., :5]          # top_k(-d, 5): ascending, earlier slot first
    val = np.take_along_axis(d, order, -1)[..., 1:]
    return float(np.maximum(0.0, np.float32(h) - val).astype(np.float64).mean())

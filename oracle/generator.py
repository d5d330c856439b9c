"""CPU ORACLE (test infrastructure) for the Dis-PU generator forward pass, inference mode.

Restates DisPU/generator.py:31-88 and the blocks of Common/ops.py it reaches:
  feature_extraction_GCN :1437-1486, dense_conv :1897-1915, get_edge_feature :1856-1877,
  duplicate_up :1152-1199 (+ gen_grid :60-76), coordinate_regressor :1089-1110,
  PointShuffle2 :1012-1087 (+ grouping :154-179, weight_net_hidden :181-191),
  PointNonLocalCell :302-346,  conv wrappers Common/tf_util.py:52-185, BN :512-531.
Hyper-parameters are the ones hard-coded in generator.py:33-44 (K=16, growth 24, dense_block 4,
use_bn False, up_ratio 4).  The TF layer arithmetic (conv / matmul / softmax / BN) is not pinned by the
reference ("parity unpinned"); this file pins it as documented in oracle/mlp_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import math
from collections import OrderedDict

import numpy as np

from . import oracle as O

K_NEIGH = 16
GROWTH = 24
DENSE_BLOCKS = 4
UP_RATIO = 4
BN_EPS = 1e-3  # tf.contrib.layers.batch_norm default epsilon


# ------------------------------------------------------------------------------------ parameters ----
def layer_shapes():
    """(name, fan-shape [kh,kw,cin,cout] or [k,cin,cout]) in graph order; names are the TF variable scopes."""
    L = []
    fe = "generator/feature_extraction_coarse/"
    L.append((fe + "layer0", (1, 1, 3, 24)))
    c_in = 24
    width = 24
    for d in range(1, DENSE_BLOCKS + 1):
        if d > 1:
            L.append((fe + "layer%d_prep" % d, (1, width, 2 * GROWTH)))
            c_in = 2 * GROWTH
        L.append((fe + "layer%d/l0" % d, (1, 1, 2 * c_in, GROWTH)))
        L.append((fe + "layer%d/l1" % d, (1, 1, GROWTH + c_in, GROWTH)))
        L.append((fe + "layer%d/l2" % d, (1, 1, 2 * GROWTH + c_in, GROWTH)))
        width += 3 * GROWTH + c_in
    assert width == 480
    L.append(("generator/upshuffle_0/conv1", (1, 1, 482, 256)))
    L.append(("generator/upshuffle_0/conv2", (1, 1, 256, 128)))
    for s, cin in (("generator/coarse_coordinate_regressor/", 128), ("refine/fine_coordinate_regressor/", 256)):
        L.append((s + "fc_layer0", (1, cin, 256)))
        L.append((s + "fc_layer1", (1, 256, 64)))
        L.append((s + "fc_layer2", (1, 64, 3)))
    ps = "refine/PointShuffle/"
    L.append((ps + "PointShuffle/conv_kv", (1, 1, 128, 128)))
    L.append((ps + "PointShuffle/conv_query", (1, 1, 128, 64)))
    L.append((ps + "PointShuffle/conv_back_project", (1, 1, 64, 256)))
    L.append((ps + "skip", (1, 134, 256)))
    L.append((ps + "conv0", (1, 1, 134, 128)))
    L.append((ps + "conv1", (1, 1, 128, 128)))
    L.append((ps + "weight_net/wconv0", (1, 1, 3, 16)))
    L.append((ps + "after_conv", (1, 128, 16, 256)))
    L.append((ps + "aggregation", (1, 256, 256)))
    return L


def init_params(seed=1234, bias_scale=0.0, bn_random=False):
    """Xavier-uniform weights (tf.contrib.layers.xavier_initializer, tf_util.py:41-45), zero biases
    (tf_util.py:104-105), BN gamma=1 beta=0 moving_mean=0 moving_var=1.  `bias_scale`/`bn_random` perturb the
    biases / BN statistics so tests exercise those code paths too.  Weights are stored [C_in, C_out]."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for name, shp in layer_shapes():
        recept = int(np.prod(shp[:-2]))
        fan_in, fan_out = recept * shp[-2], recept * shp[-1]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        cin_total = int(np.prod(shp[:-1]))
        P[name + "/weights"] = rng.uniform(-lim, lim, (cin_total, shp[-1])).astype(np.float32)
        P[name + "/biases"] = (rng.standard_normal(shp[-1]) * bias_scale).astype(np.float32)
    bn = "refine/PointShuffle/weight_net/wconv0/bn/"
    P[bn + "gamma"] = np.ones(16, np.float32)
    P[bn + "beta"] = np.zeros(16, np.float32)
    P[bn + "moving_mean"] = np.zeros(16, np.float32)
    P[bn + "moving_variance"] = np.ones(16, np.float32)
    if bn_random:
        P[bn + "gamma"] = rng.uniform(0.5, 1.5, 16).astype(np.float32)
        P[bn + "beta"] = (rng.standard_normal(16) * 0.1).astype(np.float32)
        P[bn + "moving_mean"] = (rng.standard_normal(16) * 0.1).astype(np.float32)
        P[bn + "moving_variance"] = rng.uniform(0.5, 1.5, 16).astype(np.float32)
    return P


def num_params(P):
    return int(sum(v.size for k, v in P.items() if k.endswith("weights") or k.endswith("biases")))


def bn_scale_shift(P, scope):
    """Inference BN folded to y = x*scale + shift (scale, shift evaluated in float64 then rounded)."""
    g, b = P[scope + "gamma"].astype(np.float64), P[scope + "beta"].astype(np.float64)
    mu, var = P[scope + "moving_mean"].astype(np.float64), P[scope + "moving_variance"].astype(np.float64)
    scale = g / np.sqrt(var + BN_EPS)
    return scale.astype(np.float32), (b - mu * scale).astype(np.float32)


# -------------------------------------------------------------------------------------- primitives ----
def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def linear(x, W, b=None, relu=False):
    """x[..., K] . W[K, N] (+ b) with the pinned fmaf chain (oracle/mlp_oracle.c:orc_linear)."""
    x = np.ascontiguousarray(x, np.float32)
    K, N = W.shape
    assert x.shape[-1] == K
    x2 = x.reshape(-1, K)
    y = np.empty((x2.shape[0], N), np.float32)
    W = np.ascontiguousarray(W, np.float32)
    bb = None if b is None else np.ascontiguousarray(b, np.float32)
    O.lib().orc_linear(C.c_long(x2.shape[0]), K, N, _p(x2), C.c_long(K), _p(W), _p(bb) if bb is not None else None,
                       1 if relu else 0, _p(y), C.c_long(N))
    return y.reshape(x.shape[:-1] + (N,))


def matmul_nt(a, bt):
    a, bt = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(bt, np.float32)
    b, m, k = a.shape
    n = bt.shape[1]
    c = np.empty((b, m, n), np.float32)
    O.lib().orc_matmul_nt(b, m, n, k, _p(a), _p(bt), _p(c))
    return c


def matmul_nn(a, bm):
    a, bm = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(bm, np.float32)
    lead = a.shape[:-2]
    m, k = a.shape[-2:]
    n = bm.shape[-1]
    nb = int(np.prod(lead)) if lead else 1
    c = np.empty((nb, m, n), np.float32)
    O.lib().orc_matmul_nn(C.c_long(nb), m, n, k, _p(a.reshape(nb, m, k)), _p(bm.reshape(nb, k, n)), _p(c))
    return c.reshape(lead + (m, n))


def gen_grid(up_ratio):
    """Common/ops.py:60-76 -> [up_ratio, 2]; up_ratio 4: (-.2,-.2), (.2,-.2), (-.2,.2), (.2,.2)."""
    sq = int(math.sqrt(up_ratio)) + 1
    for i in range(sq, 0, -1):
        if up_ratio % i == 0:
            num_x, num_y = i, up_ratio // i
            break
    gx = np.linspace(-0.2, 0.2, num_x, dtype=np.float32)
    gy = np.linspace(-0.2, 0.2, num_y, dtype=np.float32)
    x, y = np.meshgrid(gx, gy)
    return np.stack([x, y], -1).reshape(-1, 2).astype(np.float32)


def gather(feat, idx):
    """feat[b, idx[b, ...], :]"""
    b = feat.shape[0]
    return feat[np.arange(b).reshape((b,) + (1,) * (idx.ndim - 1)), idx]


# ----------------------------------------------------------------------------------------- blocks ----
def dense_conv(P, scope, feature, k=K_NEIGH):
    """Common/ops.py:1897-1915 (n=3, growth 24) with get_edge_feature :1856-1877."""
    _, idx2 = O.knn_point_2(k + 1, feature, feature)
    idx = idx2[:, :, 1:, 1]                                   # drop rank 0 (positional "self"), ops.py:1867
    nbr = gather(feature, idx)                                # [B,N,k,C]
    central = np.broadcast_to(feature[:, :, None, :], nbr.shape)
    y = np.concatenate([central, nbr - central], -1)
    l0 = linear(y, P[scope + "/l0/weights"], P[scope + "/l0/biases"], relu=True)
    y = np.concatenate([l0, central], -1)
    l1 = linear(y, P[scope + "/l1/weights"], P[scope + "/l1/biases"], relu=True)
    y = np.concatenate([l1, y], -1)
    l2 = linear(y, P[scope + "/l2/weights"], P[scope + "/l2/biases"], relu=False)
    y = np.concatenate([l2, y], -1)
    return y.max(axis=2), idx


def feature_extraction(P, inputs, tap=None):
    """Common/ops.py:1437-1486 -> [B,N,480]."""
    fe = "generator/feature_extraction_coarse/"
    l0 = linear(inputs, P[fe + "layer0/weights"], P[fe + "layer0/biases"], relu=False)
    out, idx = dense_conv(P, fe + "layer1", l0)
    if tap is not None:
        tap["fe_idx1"] = idx
    out = np.concatenate([out, l0], -1)
    for d in range(2, DENSE_BLOCKS + 1):
        prep = linear(out, P[fe + "layer%d_prep/weights" % d], P[fe + "layer%d_prep/biases" % d], relu=True)
        ld, idx = dense_conv(P, fe + "layer%d" % d, prep)
        if tap is not None:
            tap["fe_idx%d" % d] = idx
        out = np.concatenate([ld, out], -1)
    return out


def duplicate_up(P, feat):
    """Common/ops.py:1152-1199: copy-major tiling (output row r*N+i = copy r of point i), 2-D grid code."""
    B, N, _ = feat.shape
    grid = gen_grid(UP_RATIO)                                                  # [R,2]
    net = np.tile(feat, (1, UP_RATIO, 1))                                      # [B, R*N, C]
    g = np.repeat(grid[None, :, None, :], N, axis=2).reshape(1, UP_RATIO * N, 2)
    net = np.concatenate([net, np.broadcast_to(g, (B, UP_RATIO * N, 2))], -1)
    s = "generator/upshuffle_0/"
    net = linear(net, P[s + "conv1/weights"], P[s + "conv1/biases"], relu=True)
    return linear(net, P[s + "conv2/weights"], P[s + "conv2/biases"], relu=True)


def coordinate_regressor(P, scope, feat, is_off=False):
    """Common/ops.py:1089-1110."""
    c = linear(feat, P[scope + "fc_layer0/weights"], P[scope + "fc_layer0/biases"], relu=True)
    c = linear(c, P[scope + "fc_layer1/weights"], P[scope + "fc_layer1/biases"], relu=True)
    out = linear(c, P[scope + "fc_layer2/weights"], P[scope + "fc_layer2/biases"], relu=False)
    if is_off:
        out = (1.0 / (1.0 + np.exp(-out.astype(np.float64)))).astype(np.float32) - np.float32(0.5)
    return out


def non_local_cell(P, scope, feature):
    """Common/ops.py:302-346 with new_point = feature (npoint=1, nsample=N), bottleneck 64, mode 'dot', scaled."""
    kv = linear(feature, P[scope + "conv_kv/weights"], P[scope + "conv_kv/biases"])
    q = linear(feature, P[scope + "conv_query/weights"], P[scope + "conv_query/biases"])
    kk, vv = kv[..., :64], kv[..., 64:]
    att = matmul_nt(q, kk) / np.float32(8.0)
    att = att - att.max(-1, keepdims=True)
    e = np.exp(att.astype(np.float64))
    sm = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    out = matmul_nn(sm, vv)
    return linear(out, P[scope + "conv_back_project/weights"], P[scope + "conv_back_project/biases"], relu=True)


def point_shuffle2(P, xyz, feature, tap=None, k=K_NEIGH):
    """Common/ops.py:1012-1087 (use_knn, NL, Local; refine_point False)."""
    ps = "refine/PointShuffle/"
    idx = O.knn_batch(xyz, xyz, k).astype(np.int64)                            # ops.py:165 -> nanoflann
    if tap is not None:
        tap["ps_idx"] = idx.astype(np.int32)
    g_xyz = gather(xyz, idx)                                                   # [B,N,k,3]
    g_feat = gather(feature, idx)
    c_xyz = g_xyz - xyz[:, :, None, :]
    gf = np.concatenate([c_xyz, g_xyz, g_feat], -1)                            # 134 = 3 + 3 + 128
    nl = non_local_cell(P, ps + "PointShuffle/", feature)
    skip = linear(gf.max(axis=2), P[ps + "skip/weights"], P[ps + "skip/biases"], relu=True)
    h = linear(gf, P[ps + "conv0/weights"], P[ps + "conv0/biases"], relu=True)
    h = linear(h, P[ps + "conv1/weights"], P[ps + "conv1/biases"], relu=True)   # [B,N,k,128]
    scale, shift = bn_scale_shift(P, ps + "weight_net/wconv0/bn/")
    w = linear(c_xyz, P[ps + "weight_net/wconv0/weights"], P[ps + "weight_net/wconv0/biases"])
    w = np.maximum(w * scale + shift, np.float32(0.0))                          # [B,N,k(s),16(t)]
    hp = matmul_nn(np.ascontiguousarray(h.transpose(0, 1, 3, 2)), w)           # [B,N,128,16]: sum_s h[s,c]*w[s,t]
    B, N = xyz.shape[:2]
    a = linear(hp.reshape(B, N, 128 * 16), P[ps + "after_conv/weights"], P[ps + "after_conv/biases"], relu=True)
    a = (a + skip) + nl
    return linear(a, P[ps + "aggregation/weights"], P[ps + "aggregation/biases"], relu=True)


def generator_forward(P, inputs, tap=None):
    """DisPU/generator.py:31-88 -> (coarse [B,4N,3], fine [B,4N,3])."""
    inputs = np.ascontiguousarray(inputs, np.float32)
    feat = feature_extraction(P, inputs, tap)
    up = duplicate_up(P, feat)
    coarse = coordinate_regressor(P, "generator/coarse_coordinate_regressor/", up)
    if tap is not None:
        tap["feat480"], tap["up128"] = feat, up
    fine_feat = point_shuffle2(P, coarse, up, tap)
    off = coordinate_regressor(P, "refine/fine_coordinate_regressor/", fine_feat, is_off=True)
    if tap is not None:
        tap["fine_feat"] = fine_feat
    return coarse, coarse + off

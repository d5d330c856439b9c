"""TEST INFRASTRUCTURE -- a SECOND, independent reading of the generator graph, float64 torch-CPU.

Why it exists: no TensorFlow runs here, so oracle/generator.py (pinned fp32 fmaf chains, written around the product's needs) is
a restatement nobody could cross-check.  This file re-reads the same reference code -- DisPU/generator.py:31-88 and the blocks
of Common/ops.py it reaches -- a different way: every TF op is mapped one-to-one onto the torch LIBRARY primitive of the same
meaning (tf.nn.conv2d -> F.conv2d on NHWC->NCHW permuted tensors, tf.nn.conv1d -> F.conv1d, tf.nn.top_k(-D) -> torch.topk,
tf.gather_nd -> advanced indexing, tf.tile / tf.concat materialised as written, contrib batch_norm(is_training=False) ->
F.batch_norm(training=False, eps=1e-3), tf.nn.softmax -> F.softmax, tf.matmul -> torch.matmul), in float64, keeping the graph's
own tensor shapes ([B,N,1,C] expand_dims, [B,N,K,2C] edge tensors, [B,N,C,S] transposes).  It shares no code with
oracle/generator.py or the product: only the name -> array parameter mapping (kernels flattened [kh*kw*C_in, C_out]; the 4-D
kernel shape is re-derived here from the tensor it is applied to).  tests/test_generator_oracle.py asserts the two readings
agree to <= 2e-6 on several seeds with non-zero biases and non-trivial BN statistics: a transposed kernel, a swapped concat
order, a wrong tile order or activation would show as O(0.1).

Neighbour selection is decided in float64 here and in fp32 (GEMM-order arithmetic) in the first oracle; at a near-tie the two
can legitimately pick different neighbours.  `forward(..., knn=...)` therefore accepts the index tables of the other reading;
the test first runs WITHOUT them, and where the tables differ it checks that every differing pick is a near-tie in float64
before re-running with them.
"""
import math

import torch
import torch.nn.functional as F

F64 = torch.float64


def _t(a):
    return torch.as_tensor(a, dtype=F64)


# ---- Common/tf_util.py:52-113 (conv1d), :117-185 (conv2d), :512-531 (batch_norm_template) ------------------------
def conv2d(x, P, scope, activation=True, bn=False):
    """x [B,H,W,C_in] NHWC; kernel [kh=1, kw, C_in, C_out] VALID stride 1 -> bias_add -> (batch_norm) -> relu."""
    w = _t(P[scope + "/weights"])
    cin = x.shape[-1]
    kw = w.shape[0] // cin
    kernel = w.reshape(1, kw, cin, w.shape[1]).permute(3, 2, 0, 1)          # HWIO -> OIHW
    y = F.conv2d(x.permute(0, 3, 1, 2), kernel, bias=_t(P[scope + "/biases"]))
    if bn:
        s = scope + "/bn/"
        y = F.batch_norm(y, _t(P[s + "moving_mean"]), _t(P[s + "moving_variance"]), weight=_t(P[s + "gamma"]), bias=_t(P[s + "beta"]),
                         training=False, eps=1e-3)                          # contrib batch_norm default epsilon 0.001
    y = y.permute(0, 2, 3, 1)
    return F.relu(y) if activation else y


def conv1d(x, P, scope, activation=True):
    """x [B,L,C_in]; kernel [1, C_in, C_out]."""
    w = _t(P[scope + "/weights"])
    kernel = w.reshape(1, x.shape[-1], w.shape[1]).permute(2, 1, 0)          # WIO -> OIW
    y = F.conv1d(x.permute(0, 2, 1), kernel, bias=_t(P[scope + "/biases"])).permute(0, 2, 1)
    return F.relu(y) if activation else y


def gather_nd(params, idx):
    """tf.gather_nd(params [B,N,C], indices [..., 2] = (batch, point)) with the batch column implied: idx [B,...]."""
    b = torch.arange(params.shape[0]).reshape((-1,) + (1,) * (idx.dim() - 1)).expand_as(idx)
    return params[b, idx]


# ---- tf_ops/grouping/tf_grouping.py:61-66,95-114 ------------------------------------------------------------------
def knn_point_2(k, points, queries):
    r_a = (queries * queries).sum(dim=2, keepdim=True)
    r_b = (points * points).sum(dim=2, keepdim=True)
    m = torch.matmul(queries, points.transpose(1, 2))
    D = r_a - 2 * m + r_b.transpose(1, 2)
    # prepare_for_unique_top_k rebinds a local name (`D += ...` on a tensor argument): the caller's D is unchanged
    neg, idx = torch.topk(-D, k, dim=-1, sorted=True)
    return -neg, idx, D


# ---- Common/ops.py:1856-1915 --------------------------------------------------------------------------------------
def get_edge_feature(point_cloud, k, idx=None, tap=None, tag=None):
    if idx is None:
        _, idx, D = knn_point_2(k + 1, point_cloud, point_cloud)
        idx = idx[:, :, 1:]
        if tap is not None:
            tap[tag + "_D"] = D
    neighbors = gather_nd(point_cloud, idx)
    central = point_cloud.unsqueeze(-2).repeat(1, 1, k, 1)                   # tf.tile(expand_dims, [1,1,k,1])
    return torch.cat([central, neighbors - central], dim=-1), idx


def dense_conv(P, scope, feature, n=3, k=16, idx=None, tap=None, tag=None):
    y, idx = get_edge_feature(feature, k, idx, tap, tag)
    for i in range(n):
        if i == 0:
            y = torch.cat([conv2d(y, P, "%s/l%d" % (scope, i)), feature.unsqueeze(2).repeat(1, 1, k, 1)], dim=-1)
        elif i == n - 1:
            y = torch.cat([conv2d(y, P, "%s/l%d" % (scope, i), activation=False), y], dim=-1)
        else:
            y = torch.cat([conv2d(y, P, "%s/l%d" % (scope, i)), y], dim=-1)
    return y.max(dim=-2).values, idx


# ---- Common/ops.py:1437-1486 (dense_block = 4, growth_rate 24, use_bn False) --------------------------------------
def feature_extraction_GCN(P, inputs, scope, knn=None, tap=None):
    l0 = conv2d(inputs.unsqueeze(2), P, scope + "/layer0", activation=False).squeeze(2)
    feats, idx = dense_conv(P, scope + "/layer1", l0, idx=None if knn is None else knn[0], tap=tap, tag="fe1")
    found = [idx]
    out = torch.cat([feats, l0], dim=-1)
    for d in (2, 3, 4):
        ld = conv1d(out, P, scope + "/layer%d_prep" % d)
        ld, idx = dense_conv(P, scope + "/layer%d" % d, ld, idx=None if knn is None else knn[d - 1], tap=tap, tag="fe%d" % d)
        found.append(idx)
        out = torch.cat([ld, out], dim=-1)
    return out, found


# ---- Common/ops.py:60-76, 1152-1199 -------------------------------------------------------------------------------
def gen_grid(up_ratio):
    sqrted = int(math.sqrt(up_ratio)) + 1
    for i in reversed(range(1, sqrted + 1)):
        if up_ratio % i == 0:
            num_x, num_y = i, up_ratio // i
            break
    gx = torch.linspace(-0.2, 0.2, num_x, dtype=torch.float32)              # tf.lin_space is float32
    gy = torch.linspace(-0.2, 0.2, num_y, dtype=torch.float32)
    x, y = torch.meshgrid(gx, gy, indexing="xy")
    return torch.stack([x, y], dim=-1).reshape(-1, 2).to(F64)


def duplicate_up(P, feature, scope, up_ratio=4):
    B, N, _ = feature.shape
    net = feature.unsqueeze(2)                                               # [B,N,1,C]
    grid = gen_grid(up_ratio)                                                # [R,2]
    grid = grid.unsqueeze(0).repeat(B, 1, N)                                 # tf.tile(expand_dims(grid,0), [B,1,N]) -> [B,R,2N]
    grid = grid.reshape(B, -1, 1, 2)
    net = net.repeat(1, up_ratio, 1, 1)                                      # tf.tile(net, [1,R,1,1]): copy-major rows
    net = torch.cat([net, grid], dim=-1)
    net = conv2d(net, P, scope + "/conv1")
    net = conv2d(net, P, scope + "/conv2")
    return net.squeeze(2)


# ---- Common/ops.py:1089-1110 --------------------------------------------------------------------------------------
def coordinate_regressor(P, feature, scope, is_off=False):
    coord = conv1d(feature, P, scope + "/fc_layer0")
    coord = conv1d(coord, P, scope + "/fc_layer1")
    out = conv1d(coord, P, scope + "/fc_layer2", activation=False)
    if is_off:
        range_max = 0.5
        out = torch.sigmoid(out) * range_max * 2 - range_max
    return out


# ---- Common/ops.py:302-346 (mode 'dot', scaled, bn False) ---------------------------------------------------------
def PointNonLocalCell(P, feature, new_point, mlp, scope):
    bottleneck = mlp[0]
    B, npoint, nsample, _ = new_point.shape
    kv = conv2d(feature.unsqueeze(2), P, scope + "/conv_kv", activation=False)            # [B,ndataset,1,2*bottleneck]
    q = conv2d(new_point, P, scope + "/conv_query", activation=False).reshape(B, npoint * nsample, bottleneck)
    f1 = kv[:, :, :, :bottleneck].squeeze(2)
    f2 = kv[:, :, :, bottleneck:].squeeze(2)
    att = torch.matmul(q, f1.transpose(1, 2)) / math.sqrt(float(bottleneck))
    att = F.softmax(att, dim=-1)
    out = torch.matmul(att, f2).reshape(B, npoint, nsample, bottleneck)
    return conv2d(out, P, scope + "/conv_back_project").squeeze(1)


# ---- Common/ops.py:154-191, 1012-1087 (use_knn, NL, Local, use_bn False, refine_point False) ------------------------
def knn_query(K, src_xyz, q_xyz):
    """libs/nearest_neighbors (nanoflann L2) through Common/ops.py:110-118: K nearest by squared distance, nearest first."""
    d = ((q_xyz.unsqueeze(2) - src_xyz.unsqueeze(1)) ** 2).sum(-1)
    return torch.topk(-d, K, dim=-1, sorted=True).indices, d


def PointShuffle2(P, xyz, feature, nsample, mlp, scope, idx=None, tap=None):
    if idx is None:
        idx, d = knn_query(nsample, xyz, xyz)
        if tap is not None:
            tap["ps_D"] = d
    grouped_xyz = gather_nd(xyz, idx)
    grouped_feat = torch.cat([grouped_xyz, gather_nd(feature, idx)], dim=-1)              # grouping(use_xyz=True)
    grouped_xyz = grouped_xyz - xyz.unsqueeze(2).repeat(1, 1, nsample, 1)
    grouped_feat = torch.cat([grouped_xyz, grouped_feat], dim=-1)
    nl = PointNonLocalCell(P, feature, feature.unsqueeze(1), [max(32, feature.shape[-1] // 2), mlp[-1]], scope + "/" + scope.rsplit("/", 1)[-1])
    skip = conv1d(grouped_feat.max(dim=2).values, P, scope + "/skip")
    for i in range(len(mlp) - 1):
        grouped_feat = conv2d(grouped_feat, P, scope + "/conv%d" % i)
    weight = conv2d(grouped_xyz, P, scope + "/weight_net/wconv0", bn=True)                # weight_net_hidden(..., [nsample])
    grouped_feat = grouped_feat.permute(0, 1, 3, 2)                                       # [B,N,C,S]
    grouped_feat = torch.matmul(grouped_feat, weight)                                     # [B,N,C,S]
    grouped_feat = conv2d(grouped_feat, P, scope + "/after_conv").squeeze(2)              # kernel [1, C]: W = C, channels = S
    grouped_feat = grouped_feat + skip
    grouped_feat = grouped_feat + nl
    return xyz, conv1d(grouped_feat, P, scope + "/aggregation"), idx


# ---- DisPU/generator.py:31-88 ---------------------------------------------------------------------------------------
def forward(P, inputs, knn=None, tap=None):
    """-> (coarse [B,4N,3], fine [B,4N,3]) float64 numpy.  knn = dict(fe=[4 x [B,N,16]], ps=[B,4N,16]) overrides the selections."""
    x = _t(inputs)
    feat, fe_idx = feature_extraction_GCN(P, x, "generator/feature_extraction_coarse", None if knn is None else [_l(i) for i in knn["fe"]], tap)
    up = duplicate_up(P, feat, "generator/upshuffle_0")
    coarse = coordinate_regressor(P, up, "generator/coarse_coordinate_regressor")
    new_coarse, fine_feat, ps_idx = PointShuffle2(P, coarse, up, 16, [128, 128, 256], "refine/PointShuffle",
                                                  None if knn is None else _l(knn["ps"]), tap)
    fine = new_coarse + coordinate_regressor(P, fine_feat, "refine/fine_coordinate_regressor", is_off=True)
    if tap is not None:
        tap["fe_idx"] = [i.numpy() for i in fe_idx]
        tap["ps_idx"] = ps_idx.numpy()
    return coarse.numpy(), fine.numpy()


def _l(a):
    return torch.as_tensor(a, dtype=torch.int64)

"""CPU ORACLE (test infrastructure) for the Dis-PU TRAINING step: generator forward with is_training=True, the
reference's total loss, its gradients and one Adam update.

The reference builds this with TF1 autodiff (DisPU/model.py:68-87 loss, :158-178 optimizer); no TensorFlow is
available, so the graph is restated here on torch-CPU float64 tensors and differentiated with torch.autograd --
the same chain rule TF applies, evaluated at higher precision than the fp32 product path ("parity unpinned" at the
TF boundary, see DESIGN.md: TF's own gradient kernels cannot run here).  Conventions that matter for parity:

* neighbour indices (feature k-NN, xyz k-NN, ball query, arg-min of nn_distance) carry no gradient in TF either
  (top_k / custom ops without registered index gradients): they are computed by the fp32 numpy/C oracle and fed in.
* tf.reduce_max's gradient splits evenly between tied maxima (math_grad._MinOrMaxGrad): restated in `max_even`.
  The `central` half of an edge feature is identical for all k neighbours, so this matters.
* tf.nn.relu's gradient is (y > 0); torch.relu agrees.  tf.nn.top_k / tf.reduce_min route to the selected entries.
* BN (tf.contrib.layers.batch_norm, tf_util.py:512-531): training normalises with the biased batch variance,
  eps 1e-3, and updates moving_mean / moving_variance in place with `decay` = bn_decay = 0.95
  (generator.py:39): moving = decay * moving + (1 - decay) * batch.  With rank-4 input and
  updates_collections=None contrib.batch_norm takes its fused path (nn.fused_batch_norm), whose batch-variance
  OUTPUT -- the one fed to the moving average -- is Bessel-corrected (x M/(M-1)); the normalisation itself uses the
  biased variance.  TF-1.x library behaviour, not pinned by anything in the reference tree.
* Adam (tf.train.AdamOptimizer, model.py:178): lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);
  m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t * m / (sqrt(v) + 1e-8); b1 = opts.beta = 0.9, b2 = 0.999.

Only tests/ and tools/ import this module.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import generator as G
from . import oracle as O

DT = torch.float64
BN_DECAY = 0.95          # DisPU/generator.py:39
BN_SCOPE = "refine/PointShuffle/weight_net/wconv0/bn/"


def to_torch(P, requires_grad=True):
    Pt = OrderedDict()
    for k, v in P.items():
        t = torch.tensor(np.asarray(v), dtype=DT)
        trainable = k.endswith("/weights") or k.endswith("/biases") or k.endswith("/gamma") or k.endswith("/beta")
        Pt[k] = t.requires_grad_(requires_grad and trainable)
    return Pt


def trainable_names(P):
    return [k for k in P if k.endswith(("/weights", "/biases", "/gamma", "/beta"))]


def neighbour_indices(P, inputs):
    """All non-differentiable neighbour sets of one forward, from the fp32 oracle (BN mode does not reach them:
    the only BN layer sits after the xyz k-NN)."""
    tap = {}
    G.generator_forward(P, np.asarray(inputs, np.float32), tap)
    return {k: v.astype(np.int64) for k, v in tap.items() if k.startswith("fe_idx") or k == "ps_idx"}


class _MaxEven(torch.autograd.Function):
    """max over `dim` whose gradient is shared evenly by tied maxima (TF reduce_max)."""

    @staticmethod
    def forward(ctx, x, dim):
        y = x.max(dim=dim, keepdim=True).values
        ctx.save_for_backward(x, y)
        ctx.dim = dim
        return y.squeeze(dim)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        ind = (x == y).to(x.dtype)
        return ind / ind.sum(dim=ctx.dim, keepdim=True) * g.unsqueeze(ctx.dim), None


def max_even(x, dim):
    return _MaxEven.apply(x, dim)


def lin(Pt, scope, x, relu=False):
    y = x @ Pt[scope + "/weights"] + Pt[scope + "/biases"]
    return torch.relu(y) if relu else y


def gather(feat, idx):
    b = feat.shape[0]
    ar = torch.arange(b).reshape((b,) + (1,) * (idx.dim() - 1))
    return feat[ar, idx]


def dense_conv(Pt, scope, feature, idx):
    nbr = gather(feature, idx)
    central = feature[:, :, None, :].expand_as(nbr)
    y = torch.cat([central, nbr - central], -1)
    l0 = lin(Pt, scope + "/l0", y, True)
    y = torch.cat([l0, central], -1)
    l1 = lin(Pt, scope + "/l1", y, True)
    y = torch.cat([l1, y], -1)
    l2 = lin(Pt, scope + "/l2", y, False)
    y = torch.cat([l2, y], -1)
    return max_even(y, 2)


def feature_extraction(Pt, x, idx, tap=None):
    fe = "generator/feature_extraction_coarse/"
    l0 = lin(Pt, fe + "layer0", x)
    dc = dense_conv(Pt, fe + "layer1", l0, idx["fe_idx1"])
    if tap is not None:
        tap["dc1"] = dc
    out = torch.cat([dc, l0], -1)
    for d in range(2, G.DENSE_BLOCKS + 1):
        prep = lin(Pt, fe + "layer%d_prep" % d, out, True)
        dc = dense_conv(Pt, fe + "layer%d" % d, prep, idx["fe_idx%d" % d])
        if tap is not None:
            tap["dc%d" % d] = dc
        out = torch.cat([dc, out], -1)
    return out


def duplicate_up(Pt, feat):
    B, N, _ = feat.shape
    grid = torch.tensor(G.gen_grid(G.UP_RATIO), dtype=DT)
    net = feat.repeat(1, G.UP_RATIO, 1)
    g = grid[None, :, None, :].expand(B, G.UP_RATIO, N, 2).reshape(B, G.UP_RATIO * N, 2)
    net = torch.cat([net, g], -1)
    s = "generator/upshuffle_0/"
    return lin(Pt, s + "conv2", lin(Pt, s + "conv1", net, True), True)


def coordinate_regressor(Pt, scope, feat, is_off=False):
    c = lin(Pt, scope + "fc_layer1", lin(Pt, scope + "fc_layer0", feat, True), True)
    out = lin(Pt, scope + "fc_layer2", c)
    return torch.sigmoid(out) - 0.5 if is_off else out


def non_local_cell(Pt, scope, feature):
    kv = lin(Pt, scope + "conv_kv", feature)
    q = lin(Pt, scope + "conv_query", feature)
    kk, vv = kv[..., :64], kv[..., 64:]
    att = torch.softmax(q @ kk.transpose(1, 2) / 8.0, -1)
    return lin(Pt, scope + "conv_back_project", att @ vv, True)


def batch_norm(Pt, scope, x, is_training, bn_state):
    """x[..., C]; returns y and writes the updated moving statistics into bn_state (dict of float64 tensors)."""
    if is_training:
        flat = x.reshape(-1, x.shape[-1])
        mu = flat.mean(0)
        var = ((flat - mu) ** 2).mean(0)
        if bn_state is not None:
            with torch.no_grad():
                bn_state["moving_mean"] = BN_DECAY * Pt[scope + "moving_mean"] + (1 - BN_DECAY) * mu
                m = flat.shape[0]
                bn_state["moving_variance"] = (BN_DECAY * Pt[scope + "moving_variance"]
                                               + (1 - BN_DECAY) * var * (m / (m - 1.0)))
    else:
        mu, var = Pt[scope + "moving_mean"], Pt[scope + "moving_variance"]
    return (x - mu) / torch.sqrt(var + G.BN_EPS) * Pt[scope + "gamma"] + Pt[scope + "beta"]


def point_shuffle2(Pt, xyz, feature, idx, is_training, bn_state):
    ps = "refine/PointShuffle/"
    g_xyz = gather(xyz, idx)
    g_feat = gather(feature, idx)
    c_xyz = g_xyz - xyz[:, :, None, :]
    gf = torch.cat([c_xyz, g_xyz, g_feat], -1)
    nl = non_local_cell(Pt, ps + "PointShuffle/", feature)
    skip = lin(Pt, ps + "skip", max_even(gf, 2), True)
    h = lin(Pt, ps + "conv1", lin(Pt, ps + "conv0", gf, True), True)
    w = torch.relu(batch_norm(Pt, BN_SCOPE, lin(Pt, ps + "weight_net/wconv0", c_xyz), is_training, bn_state))
    hp = h.transpose(2, 3) @ w                                              # [B,N,128,16]
    B, N = xyz.shape[:2]
    a = lin(Pt, ps + "after_conv", hp.reshape(B, N, 128 * 16), True)
    return lin(Pt, ps + "aggregation", (a + skip) + nl, True)


def generator_forward(Pt, inputs, idx, is_training=True, bn_state=None, tap=None):
    x = torch.as_tensor(np.asarray(inputs), dtype=DT) if not torch.is_tensor(inputs) else inputs
    feat = feature_extraction(Pt, x, {k: torch.as_tensor(v) for k, v in idx.items()}, tap)
    up = duplicate_up(Pt, feat)
    coarse = coordinate_regressor(Pt, "generator/coarse_coordinate_regressor/", up)
    fine_feat = point_shuffle2(Pt, coarse, up, torch.as_tensor(idx["ps_idx"]), is_training, bn_state)
    off = coordinate_regressor(Pt, "refine/fine_coordinate_regressor/", fine_feat, is_off=True)
    if tap is not None:
        tap.update(feat480=feat, up128=up, fine_feat=fine_feat, coarse=coarse)
    return coarse, coarse + off


# ------------------------------------------------------------------------------------------- loss ----
def chamfer(pred, gt, radius):
    """loss_utils.py:45-64 with nn_distance(gt, pred): arg-min carries the gradient (tf_nndistance.py:31-37)."""
    d = ((gt[:, :, None, :] - pred[:, None, :, :]) ** 2).sum(-1)            # [B, n_gt, n_pred]
    fwd = d.min(2).values.mean(1)
    bwd = d.min(1).values.mean(1)
    return ((fwd + bwd) / radius).mean()


def repulsion(pred, nsample=20, radius=0.07, h=0.001):
    """loss_utils.py:271-298; ball-query indices from the fp32 oracle (no gradient)."""
    p32 = pred.detach().to(torch.float32).numpy()
    idx, _ = O.query_ball_point(radius, nsample, p32, p32)
    grouped = gather(pred, torch.as_tensor(idx.astype(np.int64))) - pred[:, :, None, :]
    dists = (grouped ** 2).sum(-1)
    val = torch.topk(-dists, 5, dim=-1).values[:, :, 1:]
    return torch.clamp(h + val, min=0.0).mean()


def weight_fine(epoch):
    """model.py:52-54: piecewise_constant(epoch, [10, 20, 30], [0.01, 0.1, 0.5, 1.0]) (x <= boundary -> left value)."""
    return 0.01 if epoch <= 10 else 0.1 if epoch <= 20 else 0.5 if epoch <= 30 else 1.0


def learning_rate(epoch, base=1e-3, decay_step=30, rate=0.7, clip=1e-6):
    """model.py:160-170: staircase exponential decay over epochs, clipped from below."""
    return max(base * rate ** math.floor(epoch / decay_step), clip)


def pu_loss(coarse, fine, gt, radius, epoch=0, repulsion_w=1.0, use_repulse=True):
    """model.py:75-87 -> (total, dict of terms)."""
    r = torch.as_tensor(np.asarray(radius), dtype=DT)
    cd_c = 1000.0 * chamfer(coarse, gt, r)
    cd_f = 1000.0 * chamfer(fine, gt, r)
    rep = repulsion_w * repulsion(fine) if use_repulse else torch.zeros((), dtype=DT)
    total = cd_c + weight_fine(epoch) * cd_f + rep
    return total, {"dis_coarse_cd": cd_c, "dis_fine_cd": cd_f, "repulsion_loss": rep}


def loss_and_grads(P, inputs, gt, radius, epoch=0, is_training=True, act_grads=None):
    """-> (loss float, terms, {name: grad ndarray float64}, bn_state, (coarse, fine) float64 ndarrays).
    `act_grads` (a dict) additionally receives d loss / d activation for the dense blocks' outputs dc1..dc4,
    coarse, fine and the PointShuffle2 output fine_feat (row-wise checks of the backward, see tests)."""
    Pt = to_torch(P)
    idx = neighbour_indices(P, inputs)
    bn_state, tap = {}, {}
    coarse, fine = generator_forward(Pt, inputs, idx, is_training, bn_state, tap)
    tap["fine"] = fine
    watch = ("dc1", "dc2", "dc3", "dc4", "coarse", "fine", "fine_feat")
    if act_grads is not None:
        for k in watch:
            tap[k].retain_grad()
    total, terms = pu_loss(coarse, fine, torch.as_tensor(np.asarray(gt), dtype=DT), radius, epoch)
    names = trainable_names(P)
    total.backward()
    gd = {n: (Pt[n].grad.numpy() if Pt[n].grad is not None else np.zeros(P[n].shape)) for n in names}
    if act_grads is not None:
        for k in watch:
            act_grads[k] = tap[k].grad.numpy()
    return (float(total.detach()), {k: float(v.detach()) for k, v in terms.items()}, gd,
            {k: v.numpy() for k, v in bn_state.items()}, (coarse.detach().numpy(), fine.detach().numpy()))


def adam_step(P, grads, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One tf.train.AdamOptimizer update in float64; state = {"t": int, "m": {...}, "v": {...}} (updated in place).
    Returns the new parameter dict (float64)."""
    state["t"] = t = state.get("t", 0) + 1
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    out = OrderedDict((k, np.asarray(v, np.float64)) for k, v in P.items())
    for n, g in grads.items():
        m = state.setdefault("m", {}).get(n, 0.0) * beta1 + (1 - beta1) * g
        v = state.setdefault("v", {}).get(n, 0.0) * beta2 + (1 - beta2) * g * g
        state["m"][n], state["v"][n] = m, v
        out[n] = out[n] - lr_t * m / (np.sqrt(v) + eps)
    return out

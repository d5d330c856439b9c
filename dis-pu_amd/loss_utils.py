"""Losses of the reference's training graph (Common/loss_utils.py: chamfer :45-64, hausdorff_loss :67-84,
earth_mover :170-176, get_repulsion_loss :271-298) on the hot-path ops.  Same names / arguments / return values
(scalar tensors).  chamfer and earth_mover are differentiable w.r.t. the point sets through the registered
gradients of nn_distance / match_cost, hausdorff through them and the max reductions (the reference only logs it,
DisPU/model.py:76,79); the repulsion term here is forward-only (its gradient kernel is used by train.py)."""
import torch

from . import _lib
from .tf_approxmatch import approx_match, match_cost
from .tf_grouping import query_ball_point
from .tf_nndistance import nn_distance


def _row_mean_max(x):
    b, n = x.shape
    mean = torch.empty((b,), dtype=torch.float32, device=x.device)
    mx = torch.empty((b,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dispu_row_mean_max(b, n, _lib.ptr(x.contiguous()), _lib.ptr(mean), _lib.ptr(mx),
                                             _lib.stream_ptr(x.device)), "dispu_row_mean_max")
    return mean, mx


class _RowMean(torch.autograd.Function):
    """mean over axis 1 with the HIP reduction forward; d/dx = g / n."""

    @staticmethod
    def forward(ctx, x):
        ctx.n = x.shape[1]
        return _row_mean_max(x)[0]

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.n).unsqueeze(1).expand(-1, ctx.n).contiguous()


class _RowMax(torch.autograd.Function):
    """max over axis 1 with the HIP reduction forward; the gradient goes to the maximal entries, split evenly among ties
    (tf.reduce_max's rule)."""

    @staticmethod
    def forward(ctx, x):
        mx = _row_mean_max(x)[1]
        ctx.save_for_backward(x, mx)
        return mx

    @staticmethod
    def backward(ctx, g):
        x, mx = ctx.saved_tensors
        hit = (x == mx.unsqueeze(1)).to(x.dtype)
        return hit * (g / hit.sum(1)).unsqueeze(1)


def chamfer(pred, gt, radius=1.0, forward_weight=1.0, threshold=None, return_hd=False):
    """loss_utils.py:45-64: mean_b[(fw * mean(dist gt->pred) + mean(dist pred->gt)) / radius]."""
    if threshold is not None:
        raise NotImplementedError("threshold is never set by the reference's training graph (DisPU/model.py:75-79)")
    dists_forward, _, dists_backward, _ = nn_distance(gt, pred)
    cd = forward_weight * _RowMean.apply(dists_forward) + _RowMean.apply(dists_backward)
    return (cd / radius).sum() / cd.shape[0]


def hausdorff_loss(pred, gt, radius=1.0, forward_weight=1.0, threshold=None):
    """loss_utils.py:67-84: max_b[(fw * max(dist gt->pred) + max(dist pred->gt)) / radius]."""
    if threshold is not None:
        raise NotImplementedError("threshold is never set by the reference's training graph")
    dists_forward, _, dists_backward, _ = nn_distance(gt, pred)
    hd = forward_weight * _RowMax.apply(dists_forward) + _RowMax.apply(dists_backward)
    return (hd / radius).max()


def earth_mover(pcd1, pcd2, radius=1.0):
    """loss_utils.py:170-176: mean_b(match_cost / radius / num_points); approx_match carries no gradient."""
    assert pcd1.shape[1] == pcd2.shape[1]
    num_points = float(pcd1.shape[1])
    match = approx_match(pcd1.detach(), pcd2.detach())
    cost = match_cost(pcd1, pcd2, match) / radius
    return (cost / num_points).sum() / cost.shape[0]


def get_repulsion_loss(pred, nsample=20, radius=0.07, knn=False, use_l1=False, h=0.001):
    """loss_utils.py:271-298: ball query (radius, nsample) around every point, squared distances to the grouped
    neighbours, the 4 nearest non-first ones, mean(max(0, h - d))."""
    if knn:
        raise NotImplementedError("knn=True is dead code in the reference (hard-coded (30,1024) constant, loss_utils.py:275)")
    p = pred.detach().contiguous()
    b, n, _ = p.shape
    idx, _ = query_ball_point(radius, nsample, p, p)
    if use_l1:
        h = float(h) ** 0.5 * 2
    per_point = torch.empty((b, n), dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().dispu_repulsion(b * n, n, nsample, 1 if use_l1 else 0, float(h), _lib.ptr(p), _lib.ptr(idx),
                                          _lib.ptr(per_point), _lib.stream_ptr(p.device)), "dispu_repulsion")
    return _row_mean_max(per_point)[0].sum() / (b * 4.0)

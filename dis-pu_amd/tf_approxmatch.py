"""Approximate earth-mover matching -- reference wrapper tf_ops/approxmatch/tf_approxmatch.py:13-51."""
import torch

from . import _lib
from ._util import f32, req


def approx_match(xyz1, xyz2, arith=_lib.ARITH_CONTRACT):
    """(xyz1[b,n,3] dataset, xyz2[b,m,3] query) -> match[b,m,n].   tf_approxmatch.py:13-21; no gradient (:22)."""
    xyz1, xyz2 = f32(xyz1, "xyz1"), f32(xyz2, "xyz2")
    req(xyz1.dim() == 3 and xyz1.shape[2] == 3, "ApproxMatch expects (batch_size,num_points,3) xyz1 shape")
    req(xyz2.dim() == 3 and xyz2.shape[2] == 3 and xyz2.shape[0] == xyz1.shape[0],
        "ApproxMatch expects (batch_size,num_points,3) xyz2 shape, and batch_size must match")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    L = _lib.lib()
    match = torch.empty((b, m, n), dtype=torch.float32, device=xyz1.device)
    nbytes = L.dispu_approx_match_scratch_bytes(b, n, m)
    temp = torch.empty((nbytes // 4,), dtype=torch.float32, device=xyz1.device)
    _lib.check(L.dispu_approx_match_ws(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match), _lib.ptr(temp), nbytes, int(arith),
                                       _lib.stream_ptr(xyz1.device)), "dispu_approx_match_ws")
    return match


def match_cost_grad(xyz1, xyz2, match, arith=_lib.ARITH_CONTRACT):
    """approxmatch_module.match_cost_grad(xyz1, xyz2, match) -> (grad1[b,n,3], grad2[b,m,3])   (:45-51)."""
    xyz1, xyz2, match = f32(xyz1, "xyz1"), f32(xyz2, "xyz2"), f32(match, "match")
    req(xyz1.dim() == 3 and xyz1.shape[2] == 3, "MatchCostGrad expects (batch_size,num_points,3) xyz1 shape")
    req(xyz2.dim() == 3 and xyz2.shape[2] == 3 and xyz2.shape[0] == xyz1.shape[0],
        "MatchCostGrad expects (batch_size,num_points,3) xyz2 shape, and batch_size must match")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    req(tuple(match.shape) == (b, m, n), "MatchCost expects (batch_size,#query,#dataset) match shape")
    g1 = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    g2 = torch.empty((b, m, 3), dtype=torch.float32, device=xyz1.device)
    L = _lib.lib()
    scratch = torch.empty((max(L.dispu_match_cost_grad_scratch_bytes(b, n, m) // 4, 1),), dtype=torch.float32, device=xyz1.device)
    _lib.check(L.dispu_match_cost_grad_ws(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match), _lib.ptr(g1), _lib.ptr(g2),
                                       _lib.ptr(scratch), int(arith), _lib.stream_ptr(xyz1.device)), "dispu_match_cost_grad_ws")
    return g1, g2


class _MatchCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, match, arith):
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = torch.empty((b,), dtype=torch.float32, device=xyz1.device)
        L = _lib.lib()
        scratch = torch.empty((max(L.dispu_match_cost_scratch_bytes(b, n, m) // 4, 1),), dtype=torch.float32, device=xyz1.device)
        _lib.check(L.dispu_match_cost_ws(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match), _lib.ptr(cost), _lib.ptr(scratch),
                                      int(arith), _lib.stream_ptr(xyz1.device)), "dispu_match_cost_ws")
        ctx.save_for_backward(xyz1, xyz2, match)
        ctx.arith = arith
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        xyz1, xyz2, match = ctx.saved_tensors
        g1, g2 = match_cost_grad(xyz1, xyz2, match, ctx.arith)
        gc = grad_cost.view(-1, 1, 1)
        return g1 * gc, g2 * gc, None, None


def match_cost(xyz1, xyz2, match, arith=_lib.ARITH_CONTRACT):
    """(xyz1[b,n,3], xyz2[b,m,3], match[b,m,n]) -> cost[b].   tf_approxmatch.py:29-38; gradient to xyz1/xyz2
    scaled by grad_cost, none to match (:45-51)."""
    xyz1, xyz2, match = f32(xyz1, "xyz1"), f32(xyz2, "xyz2"), f32(match, "match")
    req(xyz1.dim() == 3 and xyz1.shape[2] == 3, "MatchCost expects (batch_size,num_points,3) xyz1 shape")
    req(xyz2.dim() == 3 and xyz2.shape[2] == 3 and xyz2.shape[0] == xyz1.shape[0],
        "MatchCost expects (batch_size,num_points,3) xyz2 shape, and batch_size must match")
    req(tuple(match.shape) == (xyz1.shape[0], xyz2.shape[1], xyz1.shape[1]),
        "MatchCost expects (batch_size,#query,#dataset) match shape")
    return _MatchCost.apply(xyz1, xyz2, match, int(arith))

"""3-NN search and inverse-distance feature interpolation -- reference wrapper
tf_ops/interpolation/tf_interpolate.py:8-34 (CPU-only TF ops there; device kernels here)."""
import torch

from . import _lib
from ._util import f32, i32, req


def three_nn(xyz1, xyz2, arith=_lib.ARITH_PLAIN):
    """(xyz1[b,n,3] unknown, xyz2[b,m,3] known) -> (dist[b,n,3] SQUARED, idx[b,n,3] i32).  tf_interpolate.py:8-17."""
    xyz1, xyz2 = f32(xyz1, "xyz1"), f32(xyz2, "xyz2")
    req(xyz1.dim() == 3 and xyz1.shape[2] == 3, "ThreeNN expects (b,n,3) xyz1 shape.")
    req(xyz2.dim() == 3 and xyz2.shape[2] == 3 and xyz2.shape[0] == xyz1.shape[0], "ThreeNN expects (b,m,3) xyz2 shape.")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    _lib.check(_lib.lib().dispu_three_nn(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist), _lib.ptr(idx), int(arith),
                                         _lib.stream_ptr(xyz1.device)), "dispu_three_nn")
    return dist, idx


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        _lib.check(_lib.lib().dispu_three_interpolate(b, m, c, n, _lib.ptr(points), _lib.ptr(idx), _lib.ptr(weight),
                                                      _lib.ptr(out), _lib.stream_ptr(points.device)), "dispu_three_interpolate")
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return _three_interpolate_grad_raw(ctx.m, idx, weight, grad_out.contiguous()), None, None


def _three_interpolate_grad_raw(m, idx, weight, grad_out):
    b, n, c = grad_out.shape
    g = torch.empty((b, m, c), dtype=torch.float32, device=grad_out.device)
    _lib.check(_lib.lib().dispu_three_interpolate_grad(b, n, c, m, _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(weight),
                                                       _lib.ptr(g), _lib.stream_ptr(grad_out.device)),
               "dispu_three_interpolate_grad")
    return g


def three_interpolate(points, idx, weight):
    """(points[b,m,c], idx[b,n,3] i32, weight[b,n,3]) -> [b,n,c].   tf_interpolate.py:19-28; gradient to
    points only (:29-34)."""
    points, idx, weight = f32(points, "points"), i32(idx, "idx"), f32(weight, "weight")
    req(points.dim() == 3, "ThreeInterpolate expects (b,m,c) points shape")
    req(idx.dim() == 3 and idx.shape[0] == points.shape[0] and idx.shape[2] == 3, "ThreeInterpolate expects (b,n,3) idx shape")
    req(weight.dim() == 3 and tuple(weight.shape) == tuple(idx.shape), "ThreeInterpolate expects (b,n,3) weight shape")
    return _ThreeInterpolate.apply(points, idx, weight)


def three_interpolate_grad(points, idx, weight, grad_out):
    """interpolate_module.three_interpolate_grad(points, idx, weight, grad_out) -> [b,m,c]  (tf_interpolate.py:29-34)."""
    points, idx, weight, grad_out = f32(points, "points"), i32(idx, "idx"), f32(weight, "weight"), f32(grad_out, "grad_out")
    req(points.dim() == 3, "ThreeInterpolateGrad expects (b,m,c) points shape")
    req(idx.dim() == 3 and idx.shape[0] == points.shape[0] and idx.shape[2] == 3, "ThreeInterpolateGrad expects (b,n,3) idx shape")
    req(weight.dim() == 3 and tuple(weight.shape) == tuple(idx.shape), "ThreeInterpolateGrad expects (b,n,3) weight shape")
    req(grad_out.dim() == 3 and grad_out.shape[0] == points.shape[0] and grad_out.shape[1] == idx.shape[1]
        and grad_out.shape[2] == points.shape[2], "ThreeInterpolateGrad expects (b,n,c) grad_out shape")
    return _three_interpolate_grad_raw(points.shape[1], idx, weight, grad_out)

"""Bidirectional nearest-neighbour (Chamfer) distance -- reference wrapper
tf_ops/nn_distance/tf_nndistance.py:14-37."""
import torch

from . import _lib
from ._util import f32, i32, req


def _raw(xyz1, xyz2, arith):
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dev = xyz1.device
    d1 = torch.empty((b, n), dtype=torch.float32, device=dev)
    i1 = torch.empty((b, n), dtype=torch.int32, device=dev)
    d2 = torch.empty((b, m), dtype=torch.float32, device=dev)
    i2 = torch.empty((b, m), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().dispu_nn_distance(b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2),
                                            _lib.ptr(i2), int(arith), _lib.stream_ptr(dev)), "dispu_nn_distance")
    return d1, i1, d2, i2


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    """nn_distance_module.nn_distance_grad(...) -> (grad_xyz1[b,n,3], grad_xyz2[b,m,3])   tf_nndistance.py:31-37."""
    xyz1, xyz2 = f32(xyz1, "xyz1"), f32(xyz2, "xyz2")
    req(xyz1.dim() == 3, "NnDistanceGrad requires xyz1 be of shape (batch,#points,3)")
    req(xyz1.shape[2] == 3, "NnDistanceGrad only accepts 3d point set xyz1")
    req(xyz2.dim() == 3, "NnDistanceGrad requires xyz2 be of shape (batch,#points,3)")
    req(xyz2.shape[2] == 3, "NnDistanceGrad only accepts 3d point set xyz2")
    req(xyz2.shape[0] == xyz1.shape[0], "NnDistanceGrad expects xyz1 and xyz2 have same batch size")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    gd1, gd2, idx1, idx2 = f32(grad_dist1, "grad_dist1"), f32(grad_dist2, "grad_dist2"), i32(idx1, "idx1"), i32(idx2, "idx2")
    req(tuple(gd1.shape) == (b, n), "NnDistanceGrad requires grad_dist1 be of shape(batch,#points)")
    req(tuple(idx1.shape) == (b, n), "NnDistanceGrad requires idx1 be of shape(batch,#points)")
    req(tuple(gd2.shape) == (b, m), "NnDistanceGrad requires grad_dist2 be of shape(batch,#points)")
    req(tuple(idx2.shape) == (b, m), "NnDistanceGrad requires idx2 be of shape(batch,#points)")
    g1 = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    g2 = torch.empty((b, m, 3), dtype=torch.float32, device=xyz1.device)
    _lib.check(_lib.lib().dispu_nn_distance_grad(b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(gd1), _lib.ptr(idx1),
                                                 _lib.ptr(gd2), _lib.ptr(idx2), _lib.ptr(g1), _lib.ptr(g2),
                                                 _lib.stream_ptr(xyz1.device)), "dispu_nn_distance_grad")
    return g1, g2


class _NnDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, arith):
        d1, i1, d2, i2 = _raw(xyz1, xyz2, arith)
        ctx.save_for_backward(xyz1, xyz2, i1, i2)
        ctx.mark_non_differentiable(i1, i2)
        return d1, i1, d2, i2

    @staticmethod
    def backward(ctx, gd1, gi1, gd2, gi2):
        xyz1, xyz2, i1, i2 = ctx.saved_tensors
        g1, g2 = nn_distance_grad(xyz1, xyz2, gd1.contiguous(), i1, gd2.contiguous(), i2)
        return g1, g2, None


def nn_distance(xyz1, xyz2, arith=_lib.ARITH_CONTRACT):
    """(xyz1[b,n,3], xyz2[b,m,3]) -> (dist1[b,n], idx1[b,n] i32, dist2[b,m], idx2[b,m] i32); squared
    distances; first minimum (lowest index) wins.   tf_nndistance.py:14-24."""
    xyz1, xyz2 = f32(xyz1, "xyz1"), f32(xyz2, "xyz2")
    req(xyz1.dim() == 3, "NnDistance requires xyz1 be of shape (batch,#points,3)")
    req(xyz1.shape[2] == 3, "NnDistance only accepts 3d point set xyz1")
    req(xyz2.dim() == 3, "NnDistance requires xyz2 be of shape (batch,#points,3)")
    req(xyz2.shape[2] == 3, "NnDistance only accepts 3d point set xyz2")
    req(xyz2.shape[0] == xyz1.shape[0], "NnDistance expects xyz1 and xyz2 have same batch size")
    return _NnDistance.apply(xyz1, xyz2, int(arith))

"""Farthest point sampling / point gather -- same names, argument order and return values as the
reference wrapper tf_ops/sampling/tf_sampling.py:29-56 (torch tensors instead of TF tensors)."""
import torch

from . import _lib
from ._util import f32, i32, req


def farthest_point_sample(npoint, inp, arith=_lib.ARITH_CONTRACT):
    """(npoint:int, inp[b,n,3] f32) -> idx[b,npoint] i32.   tf_sampling.py:48-56; no gradient (:57).
    idx[:,0] == 0; ties follow the reference kernel's rule (tf_sampling_g.cu:146,158)."""
    inp = f32(inp, "inp")
    req(int(npoint) > 0, "FarthestPointSample expects positive npoint")
    req(inp.dim() == 3 and inp.shape[2] == 3, "FarthestPointSample expects (batch_size,num_points,3) inp shape")
    b, n, _ = inp.shape
    req(n > 0, "FarthestPointSample expects (batch_size,num_points,3) inp shape")
    out = torch.empty((b, int(npoint)), dtype=torch.int32, device=inp.device)
    L = _lib.lib()
    nbytes = L.dispu_fps_scratch_bytes(b, n, int(npoint))
    temp = torch.empty((nbytes // 4,), dtype=torch.float32, device=inp.device) if nbytes else None
    _lib.check(L.dispu_fps_ws(b, n, int(npoint), _lib.ptr(inp), _lib.ptr(temp), nbytes, _lib.ptr(out), int(arith),
                              _lib.stream_ptr(inp.device)), "dispu_fps_ws")
    return out


def prob_sample(inp, inpr):
    """(inp[b,ncategory] f32 weights, inpr[b,npoints] f32 uniform randoms) -> [b,npoints] i32   tf_sampling.py:12-20
    (optional op, unused by the shipped graph; no gradient :21): the index of the first cumulative weight >= r * total."""
    inp, inpr = f32(inp, "inp"), f32(inpr, "inpr")
    req(inp.dim() == 2, "ProbSample expects (batch_size,num_choices) inp shape")
    req(inpr.dim() == 2 and inpr.shape[0] == inp.shape[0], "ProbSample expects (batch_size,num_points) inpr shape")
    b, n = inp.shape
    m = inpr.shape[1]
    req(n > 0, "ProbSample expects (batch_size,num_choices) inp shape")
    out = torch.empty((b, m), dtype=torch.int32, device=inp.device)
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)      # the op's allocate_temp {b, n}: cumulative sums
    _lib.check(_lib.lib().dispu_prob_sample(b, n, m, _lib.ptr(inp), _lib.ptr(inpr), _lib.ptr(temp), _lib.ptr(out),
                                            _lib.stream_ptr(inp.device)), "dispu_prob_sample")
    return out


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
        _lib.check(_lib.lib().dispu_gather_point(b, n, m, _lib.ptr(inp), _lib.ptr(idx), _lib.ptr(out),
                                                 _lib.stream_ptr(inp.device)), "dispu_gather_point")
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, out_g):
        (idx,) = ctx.saved_tensors
        return gather_point_grad_raw(ctx.n, idx, out_g.contiguous()), None


def gather_point_grad_raw(n, idx, out_g):
    b, m = idx.shape
    inp_g = torch.empty((b, n, 3), dtype=torch.float32, device=out_g.device)
    _lib.check(_lib.lib().dispu_gather_point_grad(b, n, m, _lib.ptr(out_g), _lib.ptr(idx), _lib.ptr(inp_g),
                                                  _lib.stream_ptr(out_g.device)), "dispu_gather_point_grad")
    return inp_g


def gather_point(inp, idx):
    """(inp[b,n,3] f32, idx[b,m] i32) -> [b,m,3].   tf_sampling.py:29-37; gradient :43-47."""
    inp, idx = f32(inp, "inp"), i32(idx, "idx")
    req(inp.dim() == 3 and inp.shape[2] == 3, "GatherPoint expects (batch_size,num_points,3) inp shape")
    req(idx.dim() == 2 and idx.shape[0] == inp.shape[0], "GatherPoint expects (batch_size,num_result) idx shape")
    return _GatherPoint.apply(inp, idx)


def gather_point_grad(inp, idx, out_g):
    """sampling_module.gather_point_grad(inp, idx, out_g) -> [b,n,3]   (tf_sampling.py:43-47)."""
    inp, idx, out_g = f32(inp, "inp"), i32(idx, "idx"), f32(out_g, "out_g")
    req(inp.dim() == 3 and inp.shape[2] == 3, "GatherPointGradGpuOp expects (batch_size,num_points,3) inp")
    req(idx.dim() == 2 and idx.shape[0] == inp.shape[0], "GatherPointGradGpuOp expects (batch_size,num_result) idx shape")
    req(out_g.dim() == 3 and out_g.shape[0] == inp.shape[0] and out_g.shape[1] == idx.shape[1] and out_g.shape[2] == 3,
        "GatherPointGradGpuOp expects (batch_size,num_result,3) out_g shape")
    return gather_point_grad_raw(inp.shape[1], idx, out_g)

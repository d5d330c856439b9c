"""Validation helpers shared by the op shims.  Shape errors raise ValueError carrying the text of
the reference op's errors::InvalidArgument (tf_ops/*/tf_*.cpp) so callers see the same message."""
import torch


def req(cond, msg):
    if not cond:
        raise ValueError(msg)


def f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor on a ROCm device" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on a ROCm device (dis-pu_amd has no CPU path)" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


def i32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor on a ROCm device" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on a ROCm device (dis-pu_amd has no CPU path)" % name)
    if t.dtype != torch.int32:
        raise TypeError("%s must be int32, got %s" % (name, t.dtype))
    return t.contiguous()

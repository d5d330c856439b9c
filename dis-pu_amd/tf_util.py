"""1x1 convolution layers of the reference's Common/tf_util.py (conv1d :52-115, conv2d :120-185, inference
batch_norm_template :512-531) on MI355X: one fp32-MFMA GEMM with the bias / BatchNorm / ReLU epilogue fused.

TF keeps variables in a global graph; here parameters are passed explicitly as a dict keyed by the same variable
scopes: '<scope>/weights' [C_in, C_out] (the reference's [1,1,C_in,C_out] kernel reshaped), '<scope>/biases',
and for bn=True '<scope>/bn/{gamma,beta,moving_mean,moving_variance}'.  Only 1x1 kernels, stride 1, 'VALID'
(the only configuration the hot path uses); inference mode (is_training=False)."""
import numpy as np
import torch

from . import _lib

BN_EPS = 1e-3      # tf.contrib.layers.batch_norm default epsilon (tf_util.py:526-531 passes none)


def bn_fold(params, scope, device):
    """(scale, shift) of inference BatchNorm: y = x*scale + shift, evaluated in float64 then rounded."""
    g = np.asarray(params[scope + "/bn/gamma"], np.float64)
    b = np.asarray(params[scope + "/bn/beta"], np.float64)
    mu = np.asarray(params[scope + "/bn/moving_mean"], np.float64)
    var = np.asarray(params[scope + "/bn/moving_variance"], np.float64)
    scale = g / np.sqrt(var + BN_EPS)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(device)
    return t(scale), t(b - mu * scale)


def _dev(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)


def conv2d(inputs, num_output_channels, kernel_size=(1, 1), scope="conv2d", params=None, bn=False, is_training=False,
           activation_fn="relu", **unused):
    """inputs [..., C_in] -> [..., num_output_channels] = act(BN(inputs . W + b)).   tf_util.py:120-185"""
    if tuple(kernel_size) != (1, 1):
        raise NotImplementedError("only 1x1 kernels are on the hot path (tf_util.py:120; SURVEY A13)")
    if is_training:
        raise NotImplementedError("inference graph only (is_training=False)")
    if not (isinstance(inputs, torch.Tensor) and inputs.is_cuda and inputs.dtype == torch.float32):
        raise ValueError("conv2d expects a float32 tensor on a ROCm device")
    x = inputs.contiguous()
    cin = x.shape[-1]
    W = _dev(params[scope + "/weights"], x.device)
    b = _dev(params[scope + "/biases"], x.device)
    if tuple(W.shape) != (cin, num_output_channels):
        raise ValueError("%s/weights has shape %s, expected (%d, %d)" % (scope, tuple(W.shape), cin, num_output_channels))
    rows = x.numel() // cin
    y = torch.empty(x.shape[:-1] + (num_output_channels,), dtype=torch.float32, device=x.device)
    scale = shift = None
    if bn:
        scale, shift = bn_fold(params, scope, x.device)
    act = {"relu": 1, None: 0, "none": 0}[activation_fn]
    _lib.check(_lib.lib().dispu_linear_bn(1, rows, cin, num_output_channels, _lib.ptr(x), cin, 0, _lib.ptr(W),
                                          num_output_channels, 0, 0, _lib.ptr(b), _lib.ptr(scale), _lib.ptr(shift), act,
                                          _lib.ptr(y), num_output_channels, 0, None, 0, 0, None, 0, 0,
                                          _lib.stream_ptr(x.device)), "dispu_linear_bn")
    return y


def conv1d(inputs, num_output_channels, kernel_size=1, scope="conv1d", params=None, bn=False, is_training=False,
           activation_fn="relu", **unused):
    """tf_util.py:52-115 (kernel_size 1)."""
    if kernel_size != 1:
        raise NotImplementedError("only kernel_size 1 is on the hot path")
    return conv2d(inputs, num_output_channels, (1, 1), scope, params, bn, is_training, activation_fn)

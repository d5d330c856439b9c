"""1x1 convolution layers of the reference's Common/tf_util.py (conv1d :52-115, conv2d :120-185, inference
batch_norm_template :512-531) on MI355X: one fp32-MFMA GEMM with the bias / BatchNorm / ReLU epilogue fused.

TF keeps variables in a global graph; here parameters are passed explicitly as a dict keyed by the same variable
scopes: '<scope>/weights' [C_in, C_out] (the reference's [1,1,C_in,C_out] kernel reshaped), '<scope>/biases',
and for bn=True '<scope>/bn/{gamma,beta,moving_mean,moving_variance}'.  Only 1x1 kernels, stride 1, 'VALID'
(the only configuration the hot path uses).  is_training=False folds BatchNorm's moving statistics into the GEMM epilogue;
is_training=True (tf.contrib.layers.batch_norm, updates_collections=None, tf_util.py:512-531) normalises with the batch
statistics and updates '<scope>/bn/moving_mean|moving_variance' IN `params` (stored back as device tensors), decay
`bn_decay` (0.9 when None)."""
import numpy as np
import torch

from . import _lib

BN_EPS = 1e-3      # tf.contrib.layers.batch_norm default epsilon (tf_util.py:526-531 passes none)


def bn_fold(params, scope, device):
    """(scale, shift) of inference BatchNorm: y = x*scale + shift, evaluated in float64 then rounded."""
    host = lambda a: (a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)).astype(np.float64)
    g, b = host(params[scope + "/bn/gamma"]), host(params[scope + "/bn/beta"])
    mu, var = host(params[scope + "/bn/moving_mean"]), host(params[scope + "/bn/moving_variance"])
    scale = g / np.sqrt(var + BN_EPS)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(device)
    return t(scale), t(b - mu * scale)


def _dev(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)


def _bn_train(y, rows, c, params, scope, bn_decay, relu):
    """contrib batch_norm in training mode on y [rows, c] in place; moving statistics updated in `params`."""
    L = _lib.lib()
    dev = y.device
    gamma, beta = _dev(params[scope + "/bn/gamma"], dev), _dev(params[scope + "/bn/beta"], dev)
    mm, mv = _dev(params[scope + "/bn/moving_mean"], dev), _dev(params[scope + "/bn/moving_variance"], dev)
    if not isinstance(params[scope + "/bn/moving_mean"], torch.Tensor):
        mm, mv = mm.clone(), mv.clone()
    decay = 0.9 if bn_decay is None else float(bn_decay)
    nb = int(L.dispu_bn_scratch_bytes(rows, min(c, 64)))
    scratch = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
    # the kernel normalises up to 64 channels per launch (a divisor of 256: lanes = channels x row groups): 64-wide column
    # chunks, then the binary pieces of the remainder; channels are independent, so chunking changes nothing
    c0 = 0
    while c0 < c:
        cc = 64 if c - c0 >= 64 else 1 << ((c - c0).bit_length() - 1)
        stats = torch.empty(3 * cc, dtype=torch.float32, device=dev)
        off = lambda t: _lib.C.c_void_p(t.data_ptr() + 4 * c0)
        _lib.check(L.dispu_bn_train(rows, cc, off(y), c, off(gamma), off(beta), BN_EPS, decay, 1 if relu else 0, off(y), c,
                                    _lib.ptr(stats), off(mm), off(mv), _lib.ptr(scratch), scratch.numel() * 8,
                                    _lib.stream_ptr(dev)), "dispu_bn_train")
        c0 += cc
    params[scope + "/bn/moving_mean"], params[scope + "/bn/moving_variance"] = mm, mv
    return y


def conv2d(inputs, num_output_channels, kernel_size=(1, 1), scope="conv2d", params=None, bn=False, is_training=False,
           activation_fn="relu", bn_decay=None, **unused):
    """inputs [..., C_in] -> [..., num_output_channels] = act(BN(inputs . W + b)).   tf_util.py:120-185"""
    if tuple(kernel_size) != (1, 1):
        raise NotImplementedError("only 1x1 kernels are on the hot path (tf_util.py:120; SURVEY A13)")
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
    act = {"relu": 1, None: 0, "none": 0}[activation_fn]
    if bn and is_training:
        # batch statistics need the whole pre-activation first: GEMM + bias, then one BatchNorm(+ReLU) pass in place
        _lib.check(_lib.lib().dispu_linear_bn(1, rows, cin, num_output_channels, _lib.ptr(x), cin, 0, _lib.ptr(W),
                                              num_output_channels, 0, 0, _lib.ptr(b), None, None, 0,
                                              _lib.ptr(y), num_output_channels, 0, None, 0, 0, None, 0, 0,
                                              _lib.stream_ptr(x.device)), "dispu_linear_bn")
        _bn_train(y, rows, num_output_channels, params, scope, bn_decay, act == 1)
        return y
    scale = shift = None
    if bn:
        scale, shift = bn_fold(params, scope, x.device)
    _lib.check(_lib.lib().dispu_linear_bn(1, rows, cin, num_output_channels, _lib.ptr(x), cin, 0, _lib.ptr(W),
                                          num_output_channels, 0, 0, _lib.ptr(b), _lib.ptr(scale), _lib.ptr(shift), act,
                                          _lib.ptr(y), num_output_channels, 0, None, 0, 0, None, 0, 0,
                                          _lib.stream_ptr(x.device)), "dispu_linear_bn")
    return y


def conv1d(inputs, num_output_channels, kernel_size=1, scope="conv1d", params=None, bn=False, is_training=False,
           activation_fn="relu", bn_decay=None, **unused):
    """tf_util.py:52-115 (kernel_size 1)."""
    if kernel_size != 1:
        raise NotImplementedError("only kernel_size 1 is on the hot path")
    return conv2d(inputs, num_output_channels, (1, 1), scope, params, bn, is_training, activation_fn, bn_decay=bn_decay)

#!/usr/bin/env python3
"""Cycle stamps of edge_bwd_kernel's phases (build the library with -DEB_STAMPS: `DISPU_EXTRA_FLAGS=-DEB_STAMPS python dis-pu_amd/build.py --force`).
Prints the cycles of: weights -> LDS | gather | forward recompute | max gradient | dy2, dy1 + masks | dy0 + scatter | weight gradients."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dispu_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())
for Cc in (48, 24):
    B, n, k = 8, 256, 16
    rows = B * n
    g = torch.Generator(device=dev).manual_seed(0)
    F = torch.randn(rows, Cc, device=dev, generator=g)
    idx = torch.randint(0, n, (rows, k + 1), dtype=torch.int32, device=dev, generator=g)
    W = [torch.randn(a, 24, device=dev, generator=g) * 0.1 for a in (2 * Cc, 24 + Cc, 48 + Cc)]
    b = [torch.randn(24, device=dev, generator=g) * 0.1 for _ in range(3)]
    dOut = torch.randn(rows, 72 + Cc, device=dev, generator=g)
    dF = torch.zeros(rows, Cc, device=dev)
    dW = [torch.zeros_like(w) for w in W]
    db = [torch.zeros_like(x) for x in b]
    need = L.dispu_edge_dense_conv_grad_scratch_floats(rows, Cc)
    sc = torch.zeros(need, device=dev)
    for _ in range(3):
        _lib.check(L.dispu_edge_dense_conv_grad(rows, n, Cc, p(F), Cc, p(idx), k + 1, 1, p(W[0]), p(b[0]), p(W[1]), p(b[1]), p(W[2]), p(b[2]), p(dOut), 72 + Cc,
                                                p(dF), Cc, p(dW[0]), p(db[0]), p(dW[1]), p(db[1]), p(dW[2]), p(db[2]), p(sc), need, _lib.stream_ptr(dev)), "grad")
    torch.cuda.synchronize()
    st = sc[need - 32:].cpu().numpy().view(np.uint64)[:8].astype(np.int64)
    names = ["weights->LDS", "gather", "forward", "max grad", "dy2,dy1 + masks", "dy0 + scatter", "dW"]
    print("C =", Cc, {names[i]: int(st[i + 1] - st[i]) for i in range(7)}, "total", int(st[7] - st[0]))

#!/usr/bin/env python3
"""dispu_ps_local_grad (the local cell's backward in one launch) at the training step's shapes, against the launches it replaces."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dispu_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); st = _lib.stream_ptr(dev)
LAB = None
if sys.argv[1:]:            # lab build of csrc/ps_local_bwd.hip with the -D switches given (LB_NO_MFMA / LB_NO_DWV / LB_NO_DZ1C / LB_NO_DZ1ST / LB_NO_ATOM): wrong results, timing only
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    so = "/tmp/liblb_%d.so" % abs(hash(tuple(sys.argv[1:])))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-I" + root + "/include",
                           root + "/dis-pu_amd/csrc/ps_local_bwd.hip", root + "/dis-pu_amd/csrc/train_gemm.hip"] + sys.argv[1:] + ["-o", so])
    LAB = C.CDLL(so)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def timeit(fn, n=20, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000 / n)
    return sorted(ts)[len(ts) // 2]
for B in (8, 32):
    n, k, c, t = 1024, 16, 128, 16
    rows = B * n
    g = torch.Generator(device=dev).manual_seed(1)
    xyz = torch.rand(rows, 3, device=dev, generator=g)
    idx = torch.randint(0, n, (rows, k), device=dev, generator=g, dtype=torch.int32)
    G, A = torch.randn(rows, c, device=dev, generator=g), torch.randn(rows, c, device=dev, generator=g)
    W1 = torch.randn(c, c, device=dev, generator=g) * 0.1; W1t = W1.t().contiguous(); b1 = torch.randn(c, device=dev, generator=g) * 0.1
    Ww, bw = torch.randn(3, t, device=dev, generator=g), torch.randn(t, device=dev, generator=g)
    sc, sh = torch.ones(t, device=dev), torch.zeros(t, device=dev)
    dF = torch.randn(rows, c * t, device=dev, generator=g)
    E = lambda r, w: torch.empty((r, w), device=dev)
    h0, h1, wv, dz1, dwv, dz0, dG, dA = E(rows * k, c), E(rows * k, c), E(rows * k, t), E(rows * k, c), E(rows * k, t), E(rows * k, c), E(rows, c), E(rows, c)
    off = torch.empty((B, n + 1), dtype=torch.int32, device=dev); inv = torch.empty((B, n * k), dtype=torch.int32, device=dev)
    steps = {
        "gather_sub_relu (h0)": lambda: L.dispu_ps_gather_sub_relu(rows, n, k, c, p(idx), p(G), c, p(A), c, p(h0), c, st),
        "conv1 recompute (h1)": lambda: L.dispu_linear(1, rows * k, c, c, p(h0), c, 0, p(W1), c, 0, 0, p(b1), 1, p(h1), c, 0, None, 0, 0, None, 0, 0, st),
        "weight_net": lambda: L.dispu_ps_weight_net(rows, n, k, t, p(idx), p(xyz), p(Ww), p(bw), p(sc), p(sh), p(wv), st),
        "knn_invert": lambda: L.dispu_knn_invert(B, n, k, p(idx), p(off), p(inv), st),
        "point_matmul_grad_relu": lambda: L.dispu_ps_point_matmul_grad_relu(rows, k, c, t, p(h1), c, p(wv), p(dF), c * t, p(dz1), c, p(dwv), st),
        "conv1 dX": lambda: L.dispu_linear(1, rows * k, c, c, p(dz1), c, 0, p(W1), c, 0, 1, None, 0, p(dz0), c, 0, None, 0, 0, None, 0, 0, st),
        "conv0_gather_grad": lambda: L.dispu_ps_conv0_gather_grad(rows, n, k, c, p(idx), p(off), p(inv), p(dz0), c, p(G), c, p(A), c, p(dG), c, p(dA), c, st),
    }
    tot = 0.0
    for name, fn in (steps.items() if LAB is None else ()):
        us = timeit(fn); tot += us
        print("B=%d %-26s %7.1f us" % (B, name, us))
    fused = lambda: (L.dispu_memset_async(p(dG), 0, dG.numel() * 4, st),
                     (LAB or L).dispu_ps_local_grad(C.c_long(rows), n, p(idx), p(xyz), p(G), C.c_long(c), p(A), p(W1), p(b1), p(W1t), p(Ww), p(bw), p(sc), p(sh), p(dF), p(dz1), p(dwv), p(dG), p(dA), st))
    print("B=%d %-26s %7.1f us   (unfused launches above: %.1f us)  %s" % (B, "ps_local_grad (+ memset)", timeit(fused), tot, " ".join(sys.argv[1:])), flush=True)

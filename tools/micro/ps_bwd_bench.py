"""Stand-alone timing of the local cell's backward kernels at the 8-patch training shape (run on the GPU box): each kernel replayed
from a hipGraph of 10 launches, nothing else on the GPU (inside the step they overlap with the weight-gradient streams)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dispu_amd import _lib  # noqa: E402


def timed(fn, dev, reps=10):
    fn(_lib.stream_ptr(dev))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = _lib.stream_ptr(dev)
        for _ in range(reps):
            fn(st)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    B, M, k, c = int(os.environ.get("B", "8")), 1024, 16, 128
    rows = B * M
    torch.manual_seed(0)
    xyz = torch.rand(B, M, 3, device=dev)
    d = torch.cdist(xyz, xyz)
    idx = d.topk(k, largest=False).indices.to(torch.int32).contiguous()
    off = torch.empty(B, M + 1, dtype=torch.int32, device=dev)
    inv = torch.empty(B, M * k, dtype=torch.int32, device=dev)
    print("knn_invert %.1f us" % timed(lambda st: _lib.check(L.dispu_knn_invert(B, M, k, p(idx), p(off), p(inv), st), "inv"), dev))
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        flag = 1 if dt == torch.bfloat16 else 0
        dz0 = torch.randn(rows * k, c, device=dev).to(dt)
        Gm, Am = torch.randn(rows, c, device=dev), torch.randn(rows, c, device=dev)
        dG, dA = torch.empty(rows, c, device=dev), torch.empty(rows, c, device=dev)
        t = timed(lambda st: _lib.check(L.dispu_ps_conv0_gather_grad_s(rows, M, k, c, p(idx), p(off), p(inv), p(dz0), c, flag, p(Gm), c, p(Am), c,
                                                                       p(dG), c, p(dA), c, st), "gather"), dev)
        print("conv0_gather_grad %s %.1f us (%.0f GB/s of pair-gradient reads)" % (name, t, 2 * dz0.numel() * dz0.element_size() / t / 1e3))
        h1 = torch.randn(rows * k, c, device=dev).to(dt)
        wv = torch.randn(rows * k, 16, device=dev)
        dhp = torch.randn(rows, 2048, device=dev).to(dt)
        dz1 = torch.empty(rows * k, c, device=dev, dtype=dt)
        dwv = torch.empty(rows * k, 16, device=dev)
        t = timed(lambda st: _lib.check(L.dispu_ps_point_matmul_grad_relu_s(rows, k, c, 16, p(h1), c, p(wv), p(dhp), 2048, p(dz1), c, p(dwv), flag, st),
                                        "pmg"), dev)
        byts = h1.numel() * h1.element_size() * 2 + dhp.numel() * dhp.element_size() + wv.numel() * 8
        print("point_matmul_grad_relu %s %.1f us (%.0f GB/s)" % (name, t, byts / t / 1e3))
    W0 = torch.randn(134, c, device=dev)
    dxyz = torch.zeros(rows, 3, device=dev)
    dW0 = torch.zeros(134, c, device=dev)
    print("ps_prep_grad %.1f us" % timed(lambda st: _lib.check(L.dispu_ps_prep_grad(rows, c, p(xyz), p(W0), p(dG), c, p(dA), c, p(dxyz), p(dW0), st), "prep"), dev))
    feat = torch.relu(torch.randn(rows, c, device=dev))
    gmax = torch.zeros(rows, 144, device=dev)
    _lib.check(L.dispu_ps_skip_max(rows, M, k, c, p(idx), p(xyz), p(feat), c, p(gmax), 144, _lib.stream_ptr(dev)), "skip_max")
    dgmax = torch.randn(rows, 136, device=dev)
    dfeat = torch.zeros(rows, c, device=dev)
    print("ps_skip_max_grad %.1f us" % timed(lambda st: _lib.check(L.dispu_ps_skip_max_grad(rows, M, k, c, p(idx), p(xyz), p(feat), c, p(gmax), 144, p(dgmax), 136,
                                                                                             p(dxyz), p(dfeat), c, 1, st), "skipg"), dev))


if __name__ == "__main__":
    main()

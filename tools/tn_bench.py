"""Times dispu_linear_tn (weight-gradient product + split reduction) on the shapes of the 8-patch training step (run on the GPU box).
(The split plan is fixed in csrc/train_gemm.hip:tn_plan; rounds 3 - 4 swept it through environment switches: profiles/r03_tn_bench_variants.txt.)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dispu_amd import _lib  # noqa: E402

SHAPES = [  # (batch, M, K, N, what)
    (1, 8192, 2048, 256, "after_conv"), (1, 131072, 128, 128, "ps conv1"), (1, 8192, 256, 256, "aggregation / fc_layer0"),
    (1, 8192, 256, 64, "fc_layer1"), (1, 8192, 128, 256, "coarse fc_layer0"), (1, 8192, 256, 128, "conv2 / kv"),
    (1, 8192, 134, 256, "skip"), (1, 8192, 128, 128, "conv0 per point"), (1, 2048, 480, 256, "upshuffle conv1"),
    (8, 1024, 1024, 64, "attention dV / dK"), (1, 8192, 64, 256, "back_project"), (1, 8192, 128, 64, "query"),
]


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    st = _lib.stream_ptr(dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    tot = 0.0
    for b, M, K, N, what in SHAPES:
        X = torch.randn(b, M, K, device=dev)
        Z = torch.randn(b, M, N, device=dev)
        out = torch.zeros(b, K, N, device=dev)
        db = torch.zeros(N, device=dev)
        need = L.dispu_linear_tn_scratch_floats(b, M, K, N)
        sc = torch.empty(max(need, 1), device=dev)
        call = lambda st: _lib.check(L.dispu_linear_tn(b, M, K, N, p(X), K, M * K, p(Z), N, M * N, p(out), N, K * N, 1, p(db), p(sc), sc.numel(),
                                                       st), what)
        for _ in range(3):
            call(st)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cst = _lib.stream_ptr(dev)          # the capture stream
            for _ in range(10):
                call(cst)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        tf = 2.0 * b * M * K * N / us / 1e6
        tot += us
        print("%-26s b %d M %6d K %4d N %3d  %7.1f us  %6.1f TFLOP/s  partials %.1f MB" % (what, b, M, K, N, us, tf, need * 4 / 1e6))
    print("sum %.1f us" % tot)


if __name__ == "__main__":
    main()

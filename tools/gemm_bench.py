#!/usr/bin/env python3
"""Micro-benchmark of dispu_linear on the generator's GEMM shapes (run through gpurun).
usage: python tools/gemm_bench.py [tile codes ...]   e.g. 0 64064 64128 128128 128257   (GEMM_LIB=<path>: another build of the library)
(Rounds 2 - 4 could force a tile through an environment switch; the sweeps are in profiles/r03_gemm_tile_sweep_headline_shapes.txt.)"""
import os
import subprocess
import sys

TRAIN_SHAPES = [(8192, 256, 256), (8192, 256, 128), (8192, 128, 256), (8192, 64, 256), (8192, 256, 2048), (8192, 128, 128), (8192, 256, 64),
                (8192, 128, 64), (131072, 128, 128), (2048, 256, 480), (8192, 256, 134), (8192, 2048, 256), (8192, 134, 256), (2048, 480, 256)]        # dX / forward products of the 8-patch training step
SHAPES = TRAIN_SHAPES if os.environ.get("GEMM_SHAPES") == "train" else [(32768, 2048, 256), (524288, 128, 128), (32768, 256, 256), (8192, 480, 256), (32768, 256, 128),
          (32768, 128, 256), (32768, 134, 256), (32768, 128, 128), (32768, 64, 256), (32768, 256, 64), (8192, 360, 48),
          (32768, 144, 256), (32768, 128, 320), (8192, 120, 48), (8192, 240, 48)]


def run_one():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dispu_amd import _lib
    if os.environ.get("GEMM_LIB"):                       # A/B against another build of the library
        _lib.LIB_PATH = os.environ["GEMM_LIB"]
    L = _lib.lib()
    if os.environ.get("_GEMM_TILE"):
        L.dispu_debug_linear_tile(int(os.environ["_GEMM_TILE"]))
    dev = torch.device("cuda:0")
    out = []
    only = os.environ.get("_GEMM_ONLY")
    for (M, K, N) in (SHAPES if only is None else [SHAPES[int(only)]]):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(K, N, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        st = _lib.stream_ptr(dev)
        call = lambda: _lib.check(L.dispu_linear(1, M, K, N, x.data_ptr(), K, 0, w.data_ptr(), N, 0, 0, b.data_ptr(), 1,
                                                 y.data_ptr(), N, 0, None, 0, 0, None, 0, 0, st), "lin")
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out.append("%7dx%4dx%3d tile=%6d %8.1f us %6.1f TF/s" % (M, K, N, L.dispu_linear_tile2(1, M, K, N, 0), us, 2.0 * M * K * N / us / 1e6))
    print("\n".join(out))


if __name__ == "__main__":
    if os.environ.get("_GEMM_CHILD"):
        run_one()
    else:
        for code in (sys.argv[1:] or ["0"]):                  # tile codes to force (dispu_debug_linear_tile); 0 = the library's rule
            print("== tile %s" % code, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, _GEMM_CHILD="1", _GEMM_TILE=code))

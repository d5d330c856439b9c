#!/usr/bin/env python3
"""The 32-patch headline step with after_conv on the split-bf16 GEMM, round 4's wave-specialised kernel against round 6's streaming
kernel (dispu_debug_x3_kernel), eager two-stream launches and one stream; same-process A/B, medians of 5 loops of 50 steps."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dispu_amd import _lib, synth
from dispu_amd.generator import Generator
from dispu_amd.params import init_params
dev = torch.device("cuda:0")
gen = Generator(params=init_params(1234), device=dev)
gen.return_views = True
x = torch.from_numpy(synth.patches(32, 256, seed=2000)).to(dev)
L = _lib.lib()
def loop(n=50):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): gen(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for branches in (True, False):
    gen.branches = branches
    for mode, which in (("strict fp32", None), ("split-bf16, wave-specialised", 1), ("split-bf16, streaming", 0), ("split-bf16, wave-specialised", 1), ("split-bf16, streaming", 0)):
        gen.split_bf16 = which is not None
        if which is not None: L.dispu_debug_x3_kernel(which)
        for _ in range(30): gen(x)
        ts = sorted(loop() for _ in range(5))
        print("%-10s %-30s %.4f ms (min %.4f max %.4f)" % ("two streams" if branches else "one stream", mode, ts[2], ts[0], ts[-1]), flush=True)

#!/usr/bin/env python3
"""host time of one training step's submission: eager Python vs tape replay (no device sync inside the timed region)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dispu_amd import synth
from dispu_amd.params import init_params
from dispu_amd.train import Trainer
dev = torch.device("cuda:0")
B = 8
x, gt = synth.patch_with_gt(B, 256, 1024, seed=1)
xs, gs, rs = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev), torch.ones(B, device=dev)
for mode in ("eager", "tape"):
    tr = Trainer(params=init_params(1), device=dev)
    fn = tr.train_step if mode == "eager" else tr.train_step_taped
    for _ in range(4):
        fn(xs, gs, rs)
    torch.cuda.synchronize()
    hs, ws = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(xs, gs, rs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append(t1 - t0); ws.append(t2 - t0)
    print(mode, "host submit %.3f ms, wall %.3f ms (median of 10); tape entries %s" % (sorted(hs)[5] * 1e3, sorted(ws)[5] * 1e3,
          [len(t["tape"]) for t in tr._tapes.values()]))

#!/usr/bin/env python3
"""Which workspace buffer of the graphed training step differs from the eager step's?  (debug aid, run through gpurun)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dispu_amd import synth                      # noqa: E402
from dispu_amd.params import init_params         # noqa: E402
from dispu_amd.train import Trainer              # noqa: E402

dev = torch.device("cuda:0")
P = init_params(21)
B = 4
e, g = Trainer(params=P, device=dev), Trainer(params=P, device=dev)
rs = torch.ones(B, device=dev)
for i in range(3):
    x, gt = synth.patch_with_gt(B, 256, 1024, seed=30 + i)
    xs, gs = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev)
    for name in ("flat_p", "flat_m", "flat_v", "moving_mean", "moving_var"):
        getattr(g, name).copy_(getattr(e, name))
    g.adam_t = e.adam_t
    e.train_step(xs, gs, rs)
    g.train_step_graphed(xs, gs, rs)
    torch.cuda.synchronize()
    we, wg = e._ws[(B, 256)], g._ws[(B, 256)]
    print("== step", i)
    bad = 0
    for k in we:
        a, b = we[k], wg[k]
        if isinstance(a, torch.Tensor) and a.dtype in (torch.float32, torch.int32, torch.bfloat16):
            d = (a.float() - b.float()).abs().max().item()
            s = a.float().abs().max().item()
            if d > 1e-4 * max(s, 1e-6):
                print("  %-10s diff %.3e scale %.3e" % (k, d, s))
                bad += 1
        elif isinstance(a, list):
            for j, (aa, bb) in enumerate(zip(a, b)):
                if isinstance(aa, torch.Tensor):
                    d = (aa.float() - bb.float()).abs().max().item()
                    s = aa.float().abs().max().item()
                    if d > 1e-4 * max(s, 1e-6):
                        print("  %-10s[%d] diff %.3e scale %.3e" % (k, j, d, s))
                        bad += 1
    dg = (e.flat_g - g.flat_g).abs().max().item()
    print("  flat_g diff %.3e scale %.3e; %d buffers differ" % (dg, e.flat_g.abs().max().item(), bad))
    for name, tr in (("eager", e), ("graph", g)):
        w = tr._ws[(B, 256)]
        ref = w["dup256"].view(B, 4, 256, 256).sum(1).view(B * 256, 256)
        print("  %s: |dh256 - sum_r dup256| = %.3e (scale %.3e)" % (name, (w["dh256"] - ref).abs().max().item(), ref.abs().max().item()))

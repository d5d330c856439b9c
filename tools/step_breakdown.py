#!/usr/bin/env python3
"""Per-launch time of one generator forward (HIP events on the launch stream, eager), every launch listed.
    python tools/step_breakdown.py [B] [points per patch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dispu_amd import synth                       # noqa: E402
from dispu_amd.generator import Generator          # noqa: E402
from dispu_amd.params import init_params           # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NPT = int(sys.argv[2]) if len(sys.argv) > 2 else 256      # points per patch (1024: the second pass of 16x upsampling)
dev = torch.device("cuda:0")
gen = Generator(params=init_params(1234), device=dev)
gen.return_views = True
x = torch.from_numpy(synth.patches(B, NPT, seed=2000)).to(dev)
for _ in range(3):
    gen(x)
torch.cuda.synchronize()
reps, acc, order = 10, {}, []
for _ in range(reps):
    gen.profile = []
    gen(x)
    torch.cuda.synchronize()
    for i, (name, e0, e1) in enumerate(gen.profile):
        key = (i, name)
        if key not in acc:
            acc[key] = 0.0
            order.append(key)
        acc[key] += e0.elapsed_time(e1) * 1e3 / reps
tot = sum(acc.values())
for key in order:
    print("%3d  %-70s %8.1f us" % (key[0], key[1], acc[key]))
print("sum of launches: %.1f us" % tot)

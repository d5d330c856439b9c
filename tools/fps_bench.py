#!/usr/bin/env python3
"""farthest_point_sample timings (hipGraph replay, HIP events) at the large-cloud shapes of SURVEY 8(d)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dispu_amd.tf_sampling as S          # noqa: E402
from ops_bench import _timeit               # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for (b, n, m) in [(1, 24576, 8192), (8, 24576, 8192), (8, 8192, 2048), (32, 4097, 1024), (32, 1024, 384)]:
    for kind in ("cube", "sphere"):
        x = torch.rand(b, n, 3, device=dev, generator=g)
        if kind == "sphere":
            x = torch.randn(b, n, 3, device=dev, generator=g)
            x = x / x.norm(dim=2, keepdim=True)
        t = _timeit(lambda: S.farthest_point_sample(m, x), reps=3, warm=1)
        print(b, n, m, kind, "%.2f ms  %.0f ns/round" % (t * 1e3, t / (m - 1) * 1e9))

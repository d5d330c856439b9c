#!/usr/bin/env python3
"""approx_match / match_cost / match_cost_grad timings (HIP events) at the SURVEY 8(d) shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dispu_amd.tf_approxmatch as A          # noqa: E402
from ops_bench import _timeit                  # noqa: E402

dev = torch.device("cuda:0")
for (b, n) in [(4, 1024), (8, 1024), (16, 1024), (32, 1024), (1, 4096), (32, 4096)]:
    x1, x2 = torch.rand(b, n, 3, device=dev), torch.rand(b, n, 3, device=dev)
    t = _timeit(lambda: A.approx_match(x1, x2), reps=5, warm=2)
    mt = A.approx_match(x1, x2)
    tc = _timeit(lambda: A.match_cost(x1, x2, mt), reps=10, warm=2)
    tg = _timeit(lambda: A.match_cost_grad(x1, x2, mt), reps=10, warm=2)
    nb = b * (24 * n + 4 * n * n)
    print("(%d,%d,%d) approx_match %.1f us (%.2f T exp/s) | match_cost %.1f us (%.0f GB/s, %.1f%% HBM) | grad %.1f us"
          % (b, n, n, t * 1e6, 30.0 * b * n * n / t / 1e12, tc * 1e6, nb / tc / 1e9, nb / tc / 8e10, tg * 1e6))

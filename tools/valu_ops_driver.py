#!/usr/bin/env python3
"""One launch set of the VALU-bound ops at the shapes of the per-op table, for tools/pmc_valu_ops.sh (counters, not timings)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dispu_amd.nearest_neighbors as K      # noqa: E402
import dispu_amd.tf_grouping as G            # noqa: E402
import dispu_amd.tf_interpolate as I         # noqa: E402
import dispu_amd.tf_nndistance as D          # noqa: E402
import dispu_amd.tf_sampling as S            # noqa: E402
import dispu_amd.tf_approxmatch as A         # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
rand = lambda *s: torch.rand(*s, device=dev, generator=g)
for _ in range(3):
    x = rand(32, 1024, 3)
    K.knn_query(16, x, x)                                    # knn_xyz (32, 1024, 16): wave per query, 1024 candidates = 16 per lane
    x4 = rand(32, 4096, 3)
    K.knn_query(16, x4, x4)
    for c in (24, 48):
        f = torch.randn(32, 256, c, device=dev, generator=g)
        G.knn_point_2(17, f, f)
    G.query_ball_point(0.07, 20, x, x)
    I.three_nn(x, rand(32, 256, 3))
    D.nn_distance(x, rand(32, 1024, 3))
    S.farthest_point_sample(384, x)
    f = torch.randn(32, 1024, 48, device=dev, generator=g)
    G.knn_point_2(17, f, f)                                  # one pass over 1024 candidates (dense blocks of the second 16x pass)
    A.approx_match(x4, rand(32, 4096, 3))                    # the auction at the 16x loss shape: how many vector instructions per pair
torch.cuda.synchronize()

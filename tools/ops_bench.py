#!/usr/bin/env python3
"""Per-op micro-benchmark at the shapes of SURVEY.md 8(d): time (HIP events), algorithmic bytes (inputs once +
outputs once, the formulas of 8d) -> GB/s and fraction of the 8 TB/s HBM peak.  Most of these ops are latency /
VALU bound by construction (the table says which); group/gather are the HBM-bound family.
Run on the GPU box:  python tools/ops_bench.py > gpurun_out/ops_bench.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dispu_amd.nearest_neighbors as K          # noqa: E402
import dispu_amd.tf_approxmatch as A             # noqa: E402
import dispu_amd.tf_grouping as G                # noqa: E402
import dispu_amd.tf_interpolate as I             # noqa: E402
import dispu_amd.tf_nndistance as D              # noqa: E402
import dispu_amd.tf_sampling as S                # noqa: E402

HBM_PEAK = 8.0e12
dev = torch.device("cuda:0")


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def row(name, shape, seconds, nbytes, bound, note=""):
    return {"op": name, "shape": shape, "us": round(seconds * 1e6, 2), "algorithmic_MB": round(nbytes / 1e6, 3),
            "GBps": round(nbytes / seconds / 1e9, 1), "frac_hbm_peak": round(nbytes / seconds / HBM_PEAK, 4), "bound": bound,
            "note": note}


def main():
    out = []
    for (b, n, m) in [(256, 256, 64), (32, 1024, 384), (8, 2048, 24), (8, 24576, 8192)]:
        x = torch.rand(b, n, 3, device=dev)
        reps = 3 if n > 8192 else 20
        t = timeit(lambda: S.farthest_point_sample(m, x), reps=reps, warm=1)
        out.append(row("farthest_point_sample", [b, n, m], t, b * (12 * n + 4 * m), "latency (m-1 dependent rounds)",
                       "%.1f ns per round per cloud-block" % (t / max(m - 1, 1) * 1e9)))
    for (b, n, k) in [(32, 1024, 16), (256, 1024, 16), (32, 4096, 16)]:
        x = torch.rand(b, n, 3, device=dev)
        t = timeit(lambda: K.knn_query(k, x, x))
        out.append(row("knn_xyz (self query)", [b, n, k], t, b * (12 * n + 4 * n * k), "VALU / selection latency",
                       "%.1f M queries/s" % (b * n / t / 1e6)))
    for c in (24, 48):
        f = torch.randn(32, 256, c, device=dev)
        t = timeit(lambda: G.knn_point_2(17, f, f))
        out.append(row("knn_point_2 (feature kNN)", [32, 256, c, 17], t, 32 * (4 * 256 * c + 4 * 256 * 17), "VALU / selection latency"))
    for (b, n, ns, r) in [(32, 1024, 20, 0.07), (8, 4096, 20, 0.07)]:
        x = torch.rand(b, n, 3, device=dev)
        t = timeit(lambda: G.query_ball_point(r, ns, x, x))
        out.append(row("query_ball_point", [b, n, n, ns], t, b * (24 * n + 4 * n * ns + 4 * n), "VALU / divergence"))
    for (b, n, m, ns, c) in [(32, 512, 128, 64, 64), (32, 1024, 1024, 16, 128), (64, 1024, 1024, 20, 3), (32, 1024, 384, 64, 256)]:
        p = torch.randn(b, n, c, device=dev)
        idx = torch.randint(0, n, (b, m, ns), dtype=torch.int32, device=dev)
        t = timeit(lambda: G.group_point(p, idx))
        out.append(row("group_point", [b, n, m, ns, c], t, 4 * b * (n * c + m * ns + m * ns * c), "HBM"))
    x = torch.rand(32, 16384, 3, device=dev)
    idx = torch.randint(0, 16384, (32, 8192), dtype=torch.int32, device=dev)
    t = timeit(lambda: S.gather_point(x, idx))
    out.append(row("gather_point", [32, 16384, 8192], t, 4 * 32 * (16384 * 3 + 8192 + 8192 * 3), "HBM"))
    for (b, n, m) in [(32, 1024, 256), (32, 1024, 384)]:
        x1, x2 = torch.rand(b, n, 3, device=dev), torch.rand(b, m, 3, device=dev)
        t = timeit(lambda: I.three_nn(x1, x2))
        out.append(row("three_nn", [b, n, m], t, b * (12 * (n + m) + 24 * n), "VALU"))
    pts = torch.randn(32, 256, 256, device=dev)
    d, i3 = I.three_nn(torch.rand(32, 1024, 3, device=dev), torch.rand(32, 256, 3, device=dev))
    w = torch.rand(32, 1024, 3, device=dev)
    t = timeit(lambda: I.three_interpolate(pts, i3, w))
    out.append(row("three_interpolate", [32, 256, 256, 1024], t, 32 * (24 * 1024 + 4 * 256 * 256 + 4 * 1024 * 256), "HBM"))
    for (b, n) in [(32, 1024), (32, 4096), (1, 8192)]:
        x1, x2 = torch.rand(b, n, 3, device=dev), torch.rand(b, n, 3, device=dev)
        t = timeit(lambda: D.nn_distance(x1, x2))
        out.append(row("nn_distance (both directions)", [b, n, n], t, b * 40 * n, "VALU", "%.1f G pair-evals/s" % (2 * b * n * n / t / 1e9)))
    for (b, n) in [(4, 1024), (32, 1024), (1, 4096)]:
        x1, x2 = torch.rand(b, n, 3, device=dev), torch.rand(b, n, 3, device=dev)
        t = timeit(lambda: A.approx_match(x1, x2), reps=5, warm=1)
        out.append(row("approx_match", [b, n, n], t, b * (24 * n + 4 * n * n), "transcendental / VALU",
                       "%.1f G exp/s" % (30 * b * n * n / t / 1e9)))
        mt = A.approx_match(x1, x2)
        t = timeit(lambda: A.match_cost(x1, x2, mt), reps=5, warm=1)
        out.append(row("match_cost", [b, n, n], t, b * (24 * n + 4 * n * n), "HBM-leaning"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

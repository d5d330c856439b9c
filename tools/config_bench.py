#!/usr/bin/env python3
"""Timings of the BASELINE.json configurations other than the headline (which bench.py owns), on ONE MI355X:

  C1  one 256-point patch, 4x                      generator forward latency at B = 1 (hipGraph replay)
  C4  16x upsampling + CD + EMD, B = 32            two generator passes (256 -> 1024 -> 4096), chamfer, approx_match + match_cost
  8f-1 whole clouds (DisPU/model.py:343-381)       upsample_clouds for C = 1 / 8 / 64 clouds of 2048 points (-> 8192)
  C5  train step, 8 patches per GPU                fp32 and bf16-product steps (tools/train_bench.py measures the same)

    python tools/config_bench.py > gpurun_out/configs.json        (copy to profiles/ to keep)
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dispu_amd import loss_utils as LU                 # noqa: E402
from dispu_amd import synth                             # noqa: E402
from dispu_amd import tf_approxmatch as A              # noqa: E402
from dispu_amd import upsample as U                     # noqa: E402
from dispu_amd.generator import Generator               # noqa: E402
from dispu_amd.params import init_params                # noqa: E402
from dispu_amd.train import Trainer                     # noqa: E402
from ops_bench import _timeit                            # noqa: E402

dev = torch.device("cuda:0")


def wall(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    P = init_params(1234)
    gen = Generator(params=P, device=dev)
    gen.return_views = True
    out = {}
    c5 = {}
    xt, gtt = synth.patch_with_gt(8, 256, 1024, seed=5000)
    xt, gtt, r = torch.from_numpy(xt).to(dev), torch.from_numpy(gtt).to(dev), torch.ones(8, device=dev)
    for dt in ("f32", "bf16"):
        tr = None                                     # one trainer (and its streams) alive at a time
        tr = Trainer(params=P, device=dev, dtype=dt)
        t = wall(lambda: tr.train_step(xt, gtt, r), reps=20, warm=3)
        c5["C5 train step, 8 patches per GPU, %s" % dt] = {"ms": t * 1e3, "patches_per_s": 8 / t}

    x1 = torch.from_numpy(synth.patches(1, 256, seed=1000)).to(dev)
    t = _timeit(lambda: gen(x1), reps=20)
    out["C1 single patch 256->1024"] = {"ms": t * 1e3, "points_per_s": 1024 / t, "note": "latency of one generator forward, hipGraph replay"}

    x, gt = synth.patch_with_gt(32, 256, 4096, seed=3000)
    tx, tg = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev)
    gen.return_views = False
    t_gen = wall(lambda: U.generator_chain(gen, tx, final_ratio=16))
    _, fine = U.generator_chain(gen, tx, final_ratio=16)
    fine = fine.clone()
    t_cd = _timeit(lambda: LU.chamfer(fine, tg), reps=10)
    t_am = _timeit(lambda: A.approx_match(fine, tg), reps=3, warm=1)
    match = A.approx_match(fine, tg)
    t_mc = _timeit(lambda: A.match_cost(fine, tg, match), reps=5)
    out["C4 16x (256->1024->4096) + CD + EMD, B=32"] = {
        "generator_two_passes_ms": t_gen * 1e3, "chamfer_ms": t_cd * 1e3, "approx_match_ms": t_am * 1e3, "match_cost_ms": t_mc * 1e3,
        "total_ms": (t_gen + t_cd + t_am + t_mc) * 1e3, "points_per_s": 32 * 4096 / (t_gen + t_cd + t_am + t_mc),
        "note": "generator passes timed eagerly (wall clock around launches + sync), losses as hipGraph replays"}

    gen.return_views = False
    rng = np.random.default_rng(0)
    for C in (1, 8, 64):
        g = rng.standard_normal((C, 2048, 3))
        pcs = torch.from_numpy((g / np.linalg.norm(g, axis=2, keepdims=True)).astype(np.float32)).to(dev)
        t = wall(lambda: U.upsample_clouds(gen, pcs), reps=3, warm=1)
        out["whole clouds 2048->8192, C=%d" % C] = {"ms_per_batch": t * 1e3, "ms_per_cloud": t * 1e3 / C, "points_per_s": C * 8192 / t}

    out.update(c5)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

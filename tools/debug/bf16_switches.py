#!/usr/bin/env python3
"""bf16 8-patch training step with single Trainer attributes flipped (same process, fresh Trainer per line, 3 x 20 steps each)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dispu_amd import synth
from dispu_amd.params import init_params
from dispu_amd.train import Trainer
dev = torch.device("cuda:0")
P = init_params(1234)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
xt, gtt = synth.patch_with_gt(B, 256, 1024, seed=5000)
xt, gtt, r = torch.from_numpy(xt).to(dev), torch.from_numpy(gtt).to(dev), torch.ones(B, device=dev)
def wall(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
for name, kv in [("default", {}), ("prep_on_side", {"prep_on_side": True}), ("dw_streams=1", {"dw_streams": 1}), ("dw_streams=3", {"dw_streams": 3}),
                 ("bf16_storage off", {"bf16_storage": False}), ("tail_on_chain off", {"tail_on_chain": False}), ("default again", {})]:
    tr = None
    tr = Trainer(params=P, device=dev, dtype="bf16")
    for k, v in kv.items():
        setattr(tr, k, v)
    print("%-20s" % name, [round(wall(lambda: tr.train_step(xt, gtt, r)) * 1e3, 3) for _ in range(3)], flush=True)

"""Host time of one eager training step (Python + ctypes + event calls) against its GPU time (run on the GPU box)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dispu_amd import synth
from dispu_amd.params import init_params
from dispu_amd.train import Trainer
dev = torch.device("cuda:0")
tr = Trainer(params=init_params(1234), device=dev)
x, gt = synth.patch_with_gt(8, 256, 1024, seed=5000)
x, gt = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev)
r = torch.ones(8, device=dev)
for _ in range(5):
    tr.train_step(x, gt, r)
torch.cuda.synchronize()
# host only: every step is followed by a sync, so the host never waits inside the step
th = 0.0
for _ in range(20):
    t0 = time.perf_counter(); tr.train_step(x, gt, r); th += time.perf_counter() - t0
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    tr.train_step(x, gt, r)
torch.cuda.synchronize()
print("host time per step %.3f ms; back-to-back step %.3f ms" % (th / 20 * 1e3, (time.perf_counter() - t0) / 50 * 1e3))

import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dispu_amd import synth
from dispu_amd.params import init_params
from dispu_amd.generator import Generator
from dispu_amd.train import Trainer
dev = torch.device("cuda:0")
P = init_params(1234)
mode = sys.argv[1]
if "gen" in mode:
    gen = Generator(params=P, device=dev)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dummies = [torch.cuda.Stream(device=dev) for _ in range(skip)]
xt, gtt = synth.patch_with_gt(8, 256, 1024, seed=5000)
xt, gtt, r = torch.from_numpy(xt).to(dev), torch.from_numpy(gtt).to(dev), torch.ones(8, device=dev)
def wall(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
for dt in (("f32", "bf16") if "both" in mode else ("bf16",)):
    tr = None
    tr = Trainer(params=P, device=dev, dtype=dt)
    print(mode, skip, dt, [round(wall(lambda: tr.train_step(xt, gtt, r)) * 1e3, 4) for _ in range(3)])

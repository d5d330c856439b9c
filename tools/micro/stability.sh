cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1; done
python - <<'PY'
import torch, sys, os, time
sys.path.insert(0, os.getcwd())
from dispu_amd import synth
from dispu_amd.params import init_params
from dispu_amd.train import Trainer
dev = torch.device("cuda:0")
for dt in ("f32", "bf16"):
    tr = Trainer(params=init_params(1234), device=dev, dtype=dt)
    x, gt = synth.patch_with_gt(8, 256, 1024, seed=5000)
    x, gt = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev)
    r = torch.ones(8, device=dev)
    first = None
    t0 = time.perf_counter()
    for i in range(600):
        terms = tr.train_step(x, gt, r)
        if i % 100 == 0 or i == 599:
            v = float(terms["pu_loss"]) if isinstance(terms, dict) else float(terms[-1])
            first = v if first is None else first
            print(dt, "step", i, "pu_loss", round(v, 4), flush=True)
            assert v == v, "NaN"
    torch.cuda.synchronize()
    print(dt, "600 steps", round(time.perf_counter() - t0, 2), "s; loss", first, "->", v)
PY

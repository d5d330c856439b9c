import os, sys, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from dispu_amd import synth, upsample as U
from dispu_amd.generator import Generator
from dispu_amd.params import init_params
from ops_bench import _timeit
dev = torch.device("cuda:0")
gen = Generator(params=init_params(1234), device=dev)
x, gt = synth.patch_with_gt(32, 256, 4096, seed=3000)
tx = torch.from_numpy(x).to(dev)
gen.return_views = False
def wall(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
print("eager wall ms", wall(lambda: U.generator_chain(gen, tx, final_ratio=16)) * 1e3)
for rv in (False, True):
    gen.return_views = rv
    try:
        print("return_views", rv, "graph ms", _timeit(lambda: U.generator_chain(gen, tx, final_ratio=16), reps=3, warm=2) * 1e3)
    except Exception as e:
        print("graph failed", e)

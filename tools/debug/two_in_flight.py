"""Experiment: consecutive 32-patch steps alternating between two launch streams (two workspaces), eager launches, against one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dispu_amd import synth
from dispu_amd.generator import Generator
from dispu_amd.params import init_params

dev = torch.device("cuda:0")
P = init_params(seed=1234)
x = torch.from_numpy(synth.patches(32, 256, seed=2000)).to(dev)
nin = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gens = [Generator(params=P, device=dev) for _ in range(nin)]
for g in gens:
    g.return_views = True
streams = [torch.cuda.Stream() for _ in range(nin)]
def run(n):
    for i in range(n):
        with torch.cuda.stream(streams[i % nin]):
            gens[i % nin](x)
run(60); torch.cuda.synchronize()
res = []
for rep in range(5):
    t0 = time.perf_counter(); run(40); torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 40 * 1e3)
print("%d steps in flight: ms per 32-patch step" % nin, ["%.4f" % r for r in res])

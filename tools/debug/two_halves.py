"""Experiment: the 32-patch step as two concurrent 16-patch chains (two streams inside one hipGraph) against one 32-patch chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from dispu_amd import synth
from dispu_amd.generator import Generator
from oracle import generator as OG

dev = torch.device("cuda:0")
P = OG.init_params(seed=1234, bias_scale=0.05, bn_random=True)
x = torch.from_numpy(synth.patches(32, 256, seed=2000)).to(dev)
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gens = [Generator(params=P, device=dev) for _ in range(nsplit)]
for g in gens:
    g.return_views = True
per = 32 // nsplit
xs = [x[i * per:(i + 1) * per].contiguous() for i in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]

def step():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    outs = []
    for g, xi, s in zip(gens, xs, streams):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            outs.append(g(xi))
        e2 = torch.cuda.Event(); e2.record(s)
        cur.wait_event(e2)
    return outs

step(); torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    outs = step()
for _ in range(100):
    graph.replay()
torch.cuda.synchronize()
res = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 20 * 1e3)
print("split %d: ms per 32-patch step" % nsplit, ["%.4f" % r for r in res])
ref = Generator(params=P, device=dev)
c, f = ref(x)
torch.cuda.synchronize()
cf = torch.cat([o[1] for o in outs])
print("max |fine - single chain|", float((cf - f).abs().max()))

#!/usr/bin/env python3
"""Dump the captured training-step hipGraph as DOT and list the predecessors of chosen kernels (debug aid)."""
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dispu_amd import synth                      # noqa: E402
from dispu_amd.params import init_params         # noqa: E402
from dispu_amd.train import Trainer              # noqa: E402

dev = torch.device("cuda:0")
B = 4
g = Trainer(params=init_params(21), device=dev)
x, gt = synth.patch_with_gt(B, 256, 1024, seed=30)
xs, gs, rs = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev), torch.ones(B, device=dev)
for _ in range(2):
    g.zero_grad(); g.forward(xs); g.loss_backward(gs, rs); g.backward()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
graph.enable_debug_mode()
cap = torch.cuda.Stream()
cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    with torch.cuda.graph(graph, stream=cap):
        g.zero_grad(); g.forward(xs); g.loss_backward(gs, rs); g.backward()
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/train_graph.dot"
graph.debug_dump(out)
txt = open(out).read()
print(len(txt), "bytes of DOT")
nodes = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
print(len(nodes), "nodes", len(edges), "edges")
pred = {}
for a, b in edges:
    pred.setdefault(b, []).append(a)
for nid, lab in nodes.items():
    if "dup_sum_grad" in lab or "mlp_chain_bwd" in lab:
        print(nid, lab[:80].replace("\n", " "), "<-", [nodes.get(p, p)[:60].replace("\n", " ") for p in pred.get(nid, [])])

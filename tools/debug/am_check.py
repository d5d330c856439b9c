#!/usr/bin/env python3
"""The auction's production arithmetic (contracted d2, hardware exp) against the oracle on the reference's golden clouds and a few
synthetic shapes: largest plan-entry difference / relative EMD error per case, and the time at (32, 4096, 4096).  Used to accept or
reject arithmetic shortcuts in csrc/approxmatch.hip (round 5: factoring ratioL out of the pass-3 sum kept 4e-7; the fourth-power
exponential gave 7e-5 on the golden clouds and was rejected)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from dispu_amd import synth              # noqa: E402
import dispu_amd.tf_approxmatch as A     # noqa: E402

dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests/golden/ref_approxmatch.npz"))
cases = [("golden", z["xyz1"], z["xyz2"])]
for (b, n, m) in [(2, 256, 256), (2, 1024, 1024), (2, 300, 700), (2, 1000, 200), (1, 2048, 2048)]:
    cases.append(("%dx%dx%d" % (b, n, m), synth.patches(b, n, seed=n), synth.patches(b, m, seed=m + 1)))
for name, x1, x2 in cases:
    mo = O.approx_match(x1, x2)
    co = O.match_cost(x1, x2, mo)
    t1, t2 = torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev)
    m = A.approx_match(t1, t2)
    cost = A.match_cost(t1, t2, m).cpu().numpy()
    print("%-14s plan %.1e  EMD rel %.1e" % (name, np.abs(m.cpu().numpy() - mo).max(), (np.abs(cost - co) / np.abs(co)).max()), flush=True)
big1, big2 = torch.rand(32, 4096, 3, device=dev), torch.rand(32, 4096, 3, device=dev)
for _ in range(2):
    A.approx_match(big1, big2)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    A.approx_match(big1, big2)
torch.cuda.synchronize()
print("(32, 4096, 4096): %.0f us per call (allocations included)" % ((time.perf_counter() - t) / 5 * 1e6))

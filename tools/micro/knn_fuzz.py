"""Random shapes through the single-pass LDS k-NN kernel (1024 < n <= 4096) against the lane-per-query kernel (arith | 4): indices and
distances must be identical (run on the GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dispu_amd import nearest_neighbors as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
bad = 0
for it in range(60):
    n = int(rng.integers(1025, 4097)); m = int(rng.integers(1, 600)); k = int(rng.integers(1, 33)); b = int(rng.integers(1, 4))
    s = rng.random((b, n, 3)).astype(np.float32)
    if it % 3 == 0:                                   # grids / duplicates: many equal distances
        s = np.round(s * 6) / 6
    if it % 5 == 0:
        s[:, rng.permutation(n)[: n // 3]] = s[:, :1]
    q = np.concatenate([s[:, : m // 2], rng.random((b, m - m // 2, 3)).astype(np.float32)], 1)
    ts, tq = torch.from_numpy(s).to(dev), torch.from_numpy(q).to(dev)
    for arith in (0, 1):
        i1, d1 = K.knn_batch(ts, tq, k, return_dist=True, arith=arith)
        i2, d2 = K.knn_batch(ts, tq, k, return_dist=True, arith=arith | 4)
        if not (torch.equal(i1, i2) and torch.equal(d1, d2)):
            bad += 1
            print("MISMATCH", b, n, m, k, arith, int((i1 != i2).sum()))
print("fuzz done, mismatches:", bad)

"""Phase breakdown of knn_xyz_wave_kernel (csrc/knn*.hip built with -DKNN_STAMPS into a scratch library): cycles per wave of
workgroup (3, 1), summed over its queries.  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = "/tmp/libknn_stamps.so"
src = [ROOT + "/dis-pu_amd/csrc/" + f for f in ("knn.hip", "knn_wave.hip", "knn_general.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DKNN_STAMPS", "-shared",
                       "-I" + ROOT + "/include"] + src + sys.argv[1:] + ["-o", so])
L = C.CDLL(so)
dev = torch.device("cuda:0")
vp = C.c_void_p
b, n, k = 32, 1024, 16
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(b, n, 3, device=dev, generator=g)
idx = torch.zeros((b * n * k + 4 * 8 * 2 + 64,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    rc = L.dispu_knn_xyz(b, n, n, k, vp(x.data_ptr()), vp(x.data_ptr()), vp(idx.data_ptr()), None, 0, vp(st))
    assert rc == 0, rc
torch.cuda.synchronize()
stamps = idx[b * n * k:b * n * k + 4 * 8 * 2].cpu().numpy().view(np.uint64).reshape(4, 8)
for w in range(4):
    d, t, c, s, nq, r, pro, tot = [int(v) for v in stamps[w]]
    nf, nq = nq >> 32, nq & 0xffffffff
    print("wave %d: %d queries (%d full sorts): distances %d  threshold %d  compaction %d  rank %d  select-total %d  (cycles per query); prologue %d, whole wave %d" %
          (w, nq, nf, d // nq, t // nq, c // nq, r // nq, s // nq, pro, tot))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(50):
    L.dispu_knn_xyz(b, n, n, k, vp(x.data_ptr()), vp(x.data_ptr()), vp(idx.data_ptr()), None, 0, vp(st))
e1.record()
torch.cuda.synchronize()
print("%.2f us per launch (back to back, instrumented build)" % (e0.elapsed_time(e1) * 1000 / 50))

"""Phase breakdown of the dense-block kernels (csrc/edge.hip built with -DEDGE_STAMPS into a scratch library): cycle counts per wave
of workgroup (3, 0).  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = "/tmp/libedge_stamps.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DEDGE_STAMPS", "-shared",
                       "-I" + ROOT + "/include", ROOT + "/dis-pu_amd/csrc/edge.hip"] + sys.argv[1:] + ["-o", so])
L = C.CDLL(so)
dev = torch.device("cuda:0")
vp = C.c_void_p
for Cc in (48, 24):
    nb, n = 32, 256
    npts = nb * n
    rng = np.random.default_rng(0)
    F = torch.from_numpy(rng.standard_normal((npts, Cc)).astype(np.float32)).to(dev)
    W = [torch.from_numpy((rng.standard_normal(s) * 0.2).astype(np.float32)).to(dev) for s in
         [(2 * Cc, 24), (24,), (24 + Cc, 24), (24,), (48 + Cc, 24), (24,)]]
    k_old = 24 if Cc == 24 else 240
    ld = 72 + Cc + k_old
    Y = torch.zeros((npts * ld + 8 * 16 * 2 + 64,), device=dev)
    Wp = torch.from_numpy((rng.standard_normal((ld, 48)) * 0.1).astype(np.float32)).to(dev)
    Pb = torch.zeros((npts, 48), device=dev)
    idx = torch.zeros((npts, 17), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        rc = L.dispu_stem_block(npts, n, Cc, vp(F.data_ptr()), C.c_long(Cc), 17, 1, *[vp(w.data_ptr()) for w in W], vp(Y.data_ptr()), C.c_long(ld),
                                vp(idx.data_ptr()), vp(Wp.data_ptr()), vp(Wp.data_ptr()), k_old, vp(Pb.data_ptr()), C.c_long(48), None, None, None, None, C.c_long(0), vp(st))
        assert rc == 0, rc
    torch.cuda.synchronize()
    stamps = Y[npts * ld:npts * ld + 8 * 16 * 2].cpu().numpy().view(np.uint64).reshape(8, 16)
    names = ["gather", "l0", "l1", "l2", "epi", "groups", "stage", "phase2", "combine", "select", "total", "tail-wait", "tail-stage", "tail-mfma", "tail-loop"]
    print("C = %d (cycle counter; per wave)" % Cc)
    for w in range(8):
        print("  wave %d: " % w + "  ".join("%s %d" % (nm, int(v)) for nm, v in zip(names, stamps[w])))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(50):
        L.dispu_stem_block(npts, n, Cc, vp(F.data_ptr()), C.c_long(Cc), 17, 1, *[vp(w.data_ptr()) for w in W], vp(Y.data_ptr()), C.c_long(ld),
                           None, vp(Wp.data_ptr()), vp(Wp.data_ptr()), k_old, vp(Pb.data_ptr()), C.c_long(48), None, None, None, None, C.c_long(0), vp(st))
    e1.record()
    torch.cuda.synchronize()
    print("  %.2f us per launch (back to back)" % (e0.elapsed_time(e1) * 1000 / 50))

"""Timing lab for the wave-specialised split-bf16 GEMM (csrc/linear_bf16x3.hip built into a scratch library with the -D flags given
on the command line, e.g. -DX3_NOSPLIT / -DX3_NOMFMA) at the after_conv shape.  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = "/tmp/libx3_lab_%d.so" % abs(hash(tuple(sys.argv[1:])))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                       "-I" + ROOT + "/include", ROOT + "/dis-pu_amd/csrc/linear_bf16x3.hip"] + sys.argv[1:] + ["-o", so])
L = C.CDLL(so)
dev = torch.device("cuda:0")
vp = C.c_void_p
M, K, N = 32768, 2048, 256
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(M, K, device=dev, generator=g)
w = torch.randn(K, N, device=dev, generator=g) * 0.03
b = torch.randn(N, device=dev, generator=g)
planes = torch.empty(3 * K * N, dtype=torch.bfloat16, device=dev)
y = torch.zeros(M, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert L.dispu_bf16x3_split_weights(K, N, vp(w.data_ptr()), C.c_long(N), vp(planes.data_ptr()), vp(st)) == 0
def run():
    return L.dispu_linear_bf16x3(M, K, N, vp(x.data_ptr()), C.c_long(K), vp(planes.data_ptr()), vp(b.data_ptr()), 1, vp(y.data_ptr()), C.c_long(N),
                                 None, C.c_long(0), None, C.c_long(0), vp(st))
L.dispu_debug_x3_kernel.restype = None
outs = {}
for which, name in (((1, "wave-specialised (round 4, default)"),) if not sys.argv[1:] else ()) + ((0, "streaming (round 6)"),):
    L.dispu_debug_x3_kernel(which)
    for _ in range(5):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(5):
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    outs[which] = y.clone()
    print("%-28s" % name, " ".join(sys.argv[1:]) or "(production)", "us per launch:", " ".join("%.1f" % t for t in ts),
          "| %.0f TFLOP/s fp32-equivalent" % (2.0 * M * K * N / min(ts) / 1e6))
if not sys.argv[1:]:
    want = torch.relu(x.double() @ w.double() + b.double())
    print("max |err| vs float64:", float((y.double() - want).abs().max()), "of max", float(want.abs().max()),
          "| the two kernels bit-identical:", bool(torch.equal(outs[0], outs[1])))

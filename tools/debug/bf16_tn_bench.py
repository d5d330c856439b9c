#!/usr/bin/env python3
"""fp32 (dispu_linear_tn), bf16 (dispu_linear_tn_bf16) and streaming bf16 weight-gradient products at the training step's shapes."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dispu_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)
def timed(call):
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
for (M, K, N) in [(8192, 2048, 256), (131072, 128, 128), (8192, 256, 256), (65536, 2048, 256), (32768, 256, 128)]:
    x = torch.randn(M, K, device=dev); z = torch.randn(M, N, device=dev); out = torch.zeros(K, N, device=dev); db = torch.zeros(N, device=dev)
    xb, zb = x.to(torch.bfloat16), z.to(torch.bfloat16)
    n1 = max(L.dispu_linear_tn_scratch_floats(1, M, K, N), 1); s1 = torch.empty(n1, device=dev)
    n2 = max(L.dispu_linear_tn_bf16_scratch_floats(1, M, K, N), 1); s2 = torch.empty(n2, device=dev)
    n3 = max(L.dispu_linear_tn_bf16_stream_scratch_floats(M, K, N), 1); s3 = torch.empty(n3, device=dev)
    t1 = timed(lambda: _lib.check(L.dispu_linear_tn(1, M, K, N, x.data_ptr(), K, 0, z.data_ptr(), N, 0, out.data_ptr(), N, 0, 0, db.data_ptr(), s1.data_ptr(), n1, st), "a"))
    t2 = timed(lambda: _lib.check(L.dispu_linear_tn_bf16(1, M, K, N, x.data_ptr(), K, 0, z.data_ptr(), N, 0, out.data_ptr(), N, 0, 0, db.data_ptr(), s2.data_ptr(), n2, st), "b"))
    t3 = timed(lambda: _lib.check(L.dispu_linear_tn_bf16_stream(M, K, N, x.data_ptr(), K, z.data_ptr(), N, 0, out.data_ptr(), N, 0, db.data_ptr(), s3.data_ptr(), n3, st), "c"))
    t4 = timed(lambda: _lib.check(L.dispu_linear_tn_bf16_stream(M, K, N, xb.data_ptr(), K, zb.data_ptr(), N, 3, out.data_ptr(), N, 0, db.data_ptr(), s3.data_ptr(), n3, st), "d"))
    print("%6d x %4d x %4d  fp32 %6.1f us  bf16 %6.1f us  stream (fp32 stored) %6.1f us  stream (bf16 stored) %6.1f us  HBM floor %.1f us" % (M, K, N, t1, t2, t3, t4, (M * (K + N) * 4) / 6.5e6))

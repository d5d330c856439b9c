#!/usr/bin/env python3
"""fp32 (dispu_linear) against bf16-product (dispu_linear_bf16) forward / dX products at the training step's shapes."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dispu_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)
for (M, K, N) in [(8192, 2048, 256), (8192, 256, 2048), (131072, 128, 128), (8192, 480, 256), (8192, 256, 256), (32768, 2048, 256), (32768, 256, 2048), (65536, 2048, 256), (65536, 256, 2048), (65536, 256, 256),
                  (65536, 128, 128), (65536, 256, 128)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.05; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    wt = w.t().contiguous()
    res = []
    bt = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    _lib.check(L.dispu_bf16_pack(K, N, w.data_ptr(), N, 1, bt.data_ptr(), st), 'pack')
    for fn, tb in ((L.dispu_linear, 0), (L.dispu_linear_bf16, 0), (L.dispu_linear_bf16, 1), (None, 0)):
        if fn is None:
            call = lambda: _lib.check(L.dispu_linear_bf16_stream(M, K, N, x.data_ptr(), K, 0, bt.data_ptr(), K, b.data_ptr(), 1, y.data_ptr(), N, 0, 1, 0, st), 'stream')
        else:
            call = lambda: _lib.check(fn(1, M, K, N, x.data_ptr(), K, 0, (wt if tb else w).data_ptr(), (K if tb else N), 0, tb, b.data_ptr(), 1, y.data_ptr(), N, 0, None, 0, 0, None, 0, 0, st), "lin")
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 20)
    extra = ''
    if K >= 512:
        for nsp in (2, 4, 8):
            parts = torch.empty(nsp, M, N, device=dev)
            def call():
                _lib.check(L.dispu_linear_bf16_stream(M, K, N, x.data_ptr(), K, 0, bt.data_ptr(), K, None, 0, parts.data_ptr(), N, 0, nsp, M * N, st), 's')
                _lib.check(L.dispu_linear_splitk_finish(M, N, nsp, parts.data_ptr(), M * N, b.data_ptr(), 1, y.data_ptr(), N, st), 'f')
            for _ in range(3): call()
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize(); extra += ' split%d %.1f' % (nsp, e0.elapsed_time(e1) * 1e3 / 20)
        xb = x.to(torch.bfloat16)
    print(extra) if extra else None
    print("%6d x %4d x %4d  fp32 %7.1f us (%5.1f TF/s)  bf16 NN %7.1f us (%5.1f TF/s)  bf16 NT (W^T given) %7.1f us (%5.1f TF/s)  STREAM %7.1f us (%5.1f TF/s)  HBM floor %.1f us" % (M, K, N, res[0], 2e-6 * M * K * N / res[0], res[1], 2e-6 * M * K * N / res[1], res[2], 2e-6 * M * K * N / res[2], res[3], 2e-6 * M * K * N / res[3], (M * K + M * N + K * N) * 4 / 6.5e6))

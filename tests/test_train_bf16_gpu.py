"""Mixed-precision training step (BASELINE configs[4] names bf16; the reference is fp32-only, DisPU/model.py:215-232 /
Common/tf_util.py:52-185, so this is an extension with its OWN tolerance, stated here):

  * the bf16-product GEMMs (csrc/linear_bf16.hip: NN forward, NT input gradient, TN weight gradient) are EXACT up to
    fp32 accumulation against a float64 product of the bf16-rounded operands (<= 2e-6 of sum |a||b|) -- that pins operand
    layouts, masking, split reduction and epilogues;
  * against the fp32 operands the product error is the bf16 rounding of the operands, <= 2^-8 relative per factor;
  * a whole Trainer(dtype="bf16") step: loss terms within 2 % of the fp32 step, flat gradient cosine >= 0.97 (index
    decisions -- feature k-NN, arg-min of the Chamfer terms -- may flip at near-ties, which moves a few rows by O(1)),
    and ten Adam steps decrease the loss like the fp32 run.
"""
import numpy as np
import pytest
import torch

from oracle import generator as OG

pytestmark = pytest.mark.gpu


def N_(t):
    return t.detach().cpu().numpy()


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).bfloat16().float().numpy().astype(np.float64)


def dv(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


@pytest.mark.parametrize("M,K,N,transb,act,res", [(1000, 134, 48, 0, 1, 0), (256, 32, 128, 0, 0, 2), (4096, 2048, 256, 0, 1, 2), (513, 72, 24, 1, 0, 1),
                                                   (8192, 256, 480, 1, 0, 1), (130, 5, 7, 0, 1, 0), (64, 480, 256, 0, 0, 0)])
def test_linear_bf16_vs_float64_of_rounded_operands(dev, M, K, N, transb, act, res):
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K) if transb else (K, N)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) * 0.1
    r1, r2 = rng.standard_normal((M, N)).astype(np.float32), rng.standard_normal((M, N)).astype(np.float32)
    tx, tw, tb, t1, t2 = (dv(a, dev) for a in (x, w, bias, r1, r2))
    y = torch.full((M, N), 7.0, device=dev)
    p = lambda t: _lib.C.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(L.dispu_linear_bf16(1, M, K, N, p(tx), K, 0, p(tw), K if transb else N, 0, transb, p(tb), act, p(y), N, 0,
                                   p(t1) if res >= 1 else None, N, 0, p(t2) if res >= 2 else None, N, 0, _lib.stream_ptr(dev)), "dispu_linear_bf16")
    xr, wr = bf16_round(x), bf16_round(w)
    z = xr @ (wr.T if transb else wr) + bias
    if act:
        z = np.maximum(z, 0)
    if res >= 1:
        z = z + r1
    if res >= 2:
        z = z + r2
    bound = 2e-6 * (np.abs(xr) @ np.abs(wr.T if transb else wr)) + 1e-6 * (1 + np.abs(z))
    assert (np.abs(N_(y) - z) <= bound).all(), float(np.abs(N_(y) - z).max())
    # and against the un-rounded operands: the bf16 rounding of both factors
    zf = x.astype(np.float64) @ (w.astype(np.float64).T if transb else w.astype(np.float64))
    zb = xr @ (wr.T if transb else wr)
    assert np.abs(zb - zf).max() <= 2.0 ** -7 * (np.abs(x).astype(np.float64) @ np.abs(w.astype(np.float64).T if transb else w.astype(np.float64))).max()


def test_linear_bf16_batched_and_strided(dev):
    """the attention products of the training forward: batched, operands are column slices (ld > width)."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(1)
    b, m = 3, 160
    q = rng.standard_normal((b * m, 64)).astype(np.float32)
    kv = rng.standard_normal((b * m, 128)).astype(np.float32)
    tq, tkv = dv(q, dev), dv(kv, dev)
    s = torch.zeros((b, m, m), device=dev)
    _lib.check(L.dispu_linear_bf16(b, m, 64, m, tq.data_ptr(), 64, m * 64, tkv.data_ptr(), 128, m * 128, 1, None, 0, s.data_ptr(), m, m * m,
                                   None, 0, 0, None, 0, 0, _lib.stream_ptr(dev)), "scores")
    want = np.einsum("bqd,bkd->bqk", bf16_round(q).reshape(b, m, 64), bf16_round(kv[:, :64]).reshape(b, m, 64))
    assert np.abs(N_(s) - want).max() <= 1e-4
    o = torch.zeros((b * m, 64), device=dev)
    _lib.check(L.dispu_linear_bf16(b, m, m, 64, s.data_ptr(), m, m * m, tkv.data_ptr() + 256, 128, m * 128, 0, None, 0, o.data_ptr(), 64, m * 64,
                                   None, 0, 0, None, 0, 0, _lib.stream_ptr(dev)), "att.V")
    want2 = np.einsum("bqk,bkd->bqd", bf16_round(N_(s)), bf16_round(kv[:, 64:]).reshape(b, m, 64))
    assert np.abs(N_(o).reshape(b, m, 64) - want2).max() <= 2e-3 * np.abs(want2).max()


@pytest.mark.parametrize("M,K,N,acc", [(4096, 96, 24, 1), (131072, 120, 24, 1), (32768, 256, 128, 0), (1000, 134, 256, 1), (70000, 2048, 256, 1), (100, 7, 9, 0)])
def test_linear_tn_bf16(dev, M, K, N, acc):
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(M + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    z = (rng.standard_normal((M, N)) * 0.1).astype(np.float32)
    o0 = rng.standard_normal((K, N)).astype(np.float32)
    tx, tz, to = dv(x, dev), dv(z, dev), dv(o0, dev)
    need = L.dispu_linear_tn_bf16_scratch_floats(1, M, K, N)
    sc = torch.empty(max(need, 1), device=dev)
    db0 = rng.standard_normal(N).astype(np.float32)
    tdb = dv(db0, dev)
    _lib.check(L.dispu_linear_tn_bf16(1, M, K, N, tx.data_ptr(), K, 0, tz.data_ptr(), N, 0, to.data_ptr(), N, 0, acc, tdb.data_ptr(), sc.data_ptr(),
                                      sc.numel(), _lib.stream_ptr(dev)), "dispu_linear_tn_bf16")
    wdb = z.astype(np.float64).sum(0) + (db0 if acc else 0)                  # bias gradient: fp32 sums of the un-rounded Z
    assert np.abs(N_(tdb) - wdb).max() <= 2e-6 * np.abs(z).astype(np.float64).sum(0).max() + 1e-6
    xr, zr = bf16_round(x), bf16_round(z)
    want = xr.T @ zr + (o0 if acc else 0)
    bound = 4e-6 * (np.abs(xr).T @ np.abs(zr)) + 1e-6 * (1 + np.abs(want))
    assert (np.abs(N_(to) - want) <= bound).all(), float(np.abs(N_(to) - want).max())


@pytest.fixture(scope="module")
def steps(dev):
    from dispu_amd import synth
    from dispu_amd.train import Trainer
    P = OG.init_params(seed=1234, bias_scale=0.05, bn_random=True)
    x, gt = synth.patch_with_gt(4, 256, 1024, seed=7)
    tx, tg, r = dv(x, dev), dv(gt, dev), torch.ones(4, device=dev)
    out = {}
    for dt in ("f32", "bf16"):
        tr = Trainer(params=P, device=dev, dtype=dt)
        tr.zero_grad()
        c, f = tr.forward(tx)
        terms = tr.loss_backward(tg, r)
        tr.backward()
        torch.cuda.synchronize()
        out[dt] = dict(tr=tr, c=N_(c).copy(), f=N_(f).copy(), terms={k: float(v) for k, v in terms.items()}, g=N_(tr.flat_g).astype(np.float64))
    out["data"] = (tx, tg, r, P)
    return out


def test_bf16_step_forward_and_loss(steps):
    a, b = steps["f32"], steps["bf16"]
    assert np.isfinite(b["f"]).all()
    # coordinates: bf16 products perturb the features by ~2^-8 relative; the clouds stay within 2e-2 of the fp32 ones
    assert np.abs(b["c"] - a["c"]).max() <= 2e-2 and np.abs(b["f"] - a["f"]).max() <= 3e-2
    assert np.median(np.abs(b["f"] - a["f"])) <= 2e-3
    for k in ("dis_coarse_cd", "dis_fine_cd", "pu_loss"):
        assert abs(b["terms"][k] - a["terms"][k]) <= 2e-2 * abs(a["terms"][k]), (k, a["terms"][k], b["terms"][k])


def test_bf16_step_gradient_direction(steps):
    ga, gb = steps["f32"]["g"], steps["bf16"]["g"]
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    rel = float(np.linalg.norm(ga - gb) / np.linalg.norm(ga))
    print("bf16 vs fp32 gradient: cosine %.4f, relative L2 %.3f" % (cos, rel))
    assert cos >= 0.97, cos


def test_bf16_training_decreases_the_loss(steps, dev):
    from dispu_amd.train import Trainer
    tx, tg, r, P = steps["data"]
    hist = {}
    for dt in ("f32", "bf16"):
        tr = Trainer(params=P, device=dev, dtype=dt)
        hist[dt] = [float(tr.train_step(tx, tg, r)["pu_loss"]) for _ in range(10)]
    assert hist["bf16"][-1] < 0.9 * hist["bf16"][0], hist["bf16"]
    assert abs(hist["bf16"][-1] - hist["f32"][-1]) <= 0.15 * hist["f32"][-1], (hist["f32"][-1], hist["bf16"][-1])


# ---------------------------------------------------------------------------- round 3: bf16 activation STORAGE ----
@pytest.mark.parametrize("M,K,N,transb,xb,yb", [(4096, 128, 128, 0, 1, 1), (8192, 256, 2048, 1, 0, 1), (1000, 128, 128, 1, 1, 1), (2048, 128, 64, 0, 1, 0),
                                                (777, 100, 52, 0, 1, 1)])
def test_linear_bf16_with_bf16_stored_tensors(dev, M, K, N, transb, xb, yb):
    """dispu_linear_bf16s: X read from / Y written to bf16 tensors.  A bf16-stored X is exactly what the fp32-storage kernel rounds its
    operand to, so the product equals dispu_linear_bf16 on the widened tensor bit for bit; a bf16 Y is that result rounded once."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(M + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K) if transb else (K, N)) * 0.1).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    p = lambda t: _lib.C.c_void_p(t.data_ptr()) if t is not None else None
    tx32 = dv(x, dev)
    txb = tx32.bfloat16()
    txw = txb.float().contiguous()
    tw, tb = dv(w, dev), dv(bias, dev)
    ref = torch.empty((M, N), device=dev)
    _lib.check(L.dispu_linear_bf16(1, M, K, N, p(txw if xb else tx32), K, 0, p(tw), K if transb else N, 0, transb, p(tb), 1, p(ref), N, 0,
                                   None, 0, 0, None, 0, 0, _lib.stream_ptr(dev)), "dispu_linear_bf16")
    y = torch.empty((M, N), dtype=torch.bfloat16 if yb else torch.float32, device=dev)
    _lib.check(L.dispu_linear_bf16s(1, M, K, N, p(txb if xb else tx32), K, 0, p(tw), K if transb else N, 0, transb, p(tb), 1, p(y), N, 0, None, 0, 0,
                                    (1 if xb else 0) | (4 if yb else 0), _lib.stream_ptr(dev)), "dispu_linear_bf16s")
    want = ref.bfloat16() if yb else ref
    assert torch.equal(y, want)


@pytest.mark.parametrize("M,K,N,xb,zb", [(131072, 128, 128, 1, 1), (8192, 2048, 256, 0, 0), (5000, 128, 128, 1, 0), (3000, 100, 24, 0, 1)])
def test_linear_tn_bf16_with_bf16_stored_operands(dev, M, K, N, xb, zb):
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    z = rng.standard_normal((M, N)).astype(np.float32)
    p = lambda t: _lib.C.c_void_p(t.data_ptr()) if t is not None else None
    tx, tz = dv(x, dev), dv(z, dev)
    txb, tzb = tx.bfloat16(), tz.bfloat16()
    txw, tzw = txb.float().contiguous(), tzb.float().contiguous()          # widened copies (kept alive: the launches are asynchronous)
    need = L.dispu_linear_tn_bf16_scratch_floats(1, M, K, N)
    sc = torch.empty(max(need, 1), device=dev)
    outs = []
    for stored in (False, True):
        out, db = torch.zeros((K, N), device=dev), torch.zeros(N, device=dev)
        if stored:
            _lib.check(L.dispu_linear_tn_bf16s(1, M, K, N, p(txb if xb else tx), K, 0, p(tzb if zb else tz), N, 0, p(out), N, 0, 0, p(db), p(sc), need,
                                               (1 if xb else 0) | (2 if zb else 0), _lib.stream_ptr(dev)), "tn_bf16s")
        else:
            _lib.check(L.dispu_linear_tn_bf16(1, M, K, N, p(txw if xb else tx), K, 0, p(tzw if zb else tz), N, 0,
                                              p(out), N, 0, 0, p(db), p(sc), need, _lib.stream_ptr(dev)), "tn_bf16")
        outs.append((out, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_pair_tensor_kernels_with_bf16_storage(dev):
    """gather_sub_relu / point_matmul_grad_relu / conv0_gather_grad on bf16-stored pair tensors == the fp32-storage kernels on the
    widened tensors (outputs rounded once where they are stored as bf16)."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    B, n, k, c = 2, 256, 16, 128
    rows = B * n
    p = lambda t: _lib.C.c_void_p(t.data_ptr()) if t is not None else None
    st = _lib.stream_ptr(dev)
    idx = torch.from_numpy(rng.integers(0, n, (rows, k)).astype(np.int32)).to(dev)
    G, A = dv(rng.standard_normal((rows, c)), dev), dv(rng.standard_normal((rows, c)), dev)
    h32 = torch.empty((rows * k, c), device=dev)
    hb = torch.empty((rows * k, c), dtype=torch.bfloat16, device=dev)
    _lib.check(L.dispu_ps_gather_sub_relu(rows, n, k, c, p(idx), p(G), c, p(A), c, p(h32), c, st), "gather")
    _lib.check(L.dispu_ps_gather_sub_relu_bf16(rows, n, k, c, p(idx), p(G), c, p(A), c, p(hb), c, st), "gather_bf16")
    assert torch.equal(hb, h32.bfloat16())
    # point_matmul_grad_relu
    h1b = torch.relu(dv(rng.standard_normal((rows * k, c)), dev)).bfloat16()
    dob = dv(rng.standard_normal((rows, c * 16)), dev).bfloat16()
    wv = dv(rng.standard_normal((rows * k, 16)), dev)
    dz_b = torch.empty((rows * k, c), dtype=torch.bfloat16, device=dev)
    dz_f, dwv_b, dwv_f = torch.empty((rows * k, c), device=dev), torch.empty((rows * k, 16), device=dev), torch.empty((rows * k, 16), device=dev)
    _lib.check(L.dispu_ps_point_matmul_grad_relu_s(rows, k, c, 16, p(h1b), c, p(wv), p(dob), c * 16, p(dz_b), c, p(dwv_b), 1, st), "pmg bf16")
    h1f, dof = h1b.float().contiguous(), dob.float().contiguous()
    _lib.check(L.dispu_ps_point_matmul_grad_relu_s(rows, k, c, 16, p(h1f), c, p(wv), p(dof), c * 16, p(dz_f), c, p(dwv_f), 0, st), "pmg f32")
    assert torch.equal(dz_b, dz_f.bfloat16()) and torch.equal(dwv_b, dwv_f)
    # conv0 gather
    off = torch.empty((B, n + 1), dtype=torch.int32, device=dev)
    inv = torch.empty((B, n * k), dtype=torch.int32, device=dev)
    _lib.check(L.dispu_knn_invert(B, n, k, p(idx), p(off), p(inv), st), "invert")
    res = []
    for t, flag in ((dz_b, 1), (dz_b.float().contiguous(), 0)):
        dG, dA = torch.empty((rows, c), device=dev), torch.empty((rows, c), device=dev)
        _lib.check(L.dispu_ps_conv0_gather_grad_s(rows, n, k, c, p(idx), p(off), p(inv), p(t), c, flag, p(G), c, p(A), c, p(dG), c, p(dA), c, st), "gather grad")
        res.append((dG, dA))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,act,ybf", [(8192, 2048, 256, 1, 0), (256, 64, 128, 0, 0), (1024, 256, 2048, 0, 1), (128, 32, 128, 1, 0), (384, 96, 384, 0, 0)])
def test_streaming_bf16_product_equals_the_register_staged_one(M, K, N, act, ybf):
    """dispu_linear_bf16_stream (operands by DMA, swizzled LDS image, weights packed as bf16 [N][K]) against dispu_linear_bf16 on the same
    operands: the same roundings and the same ascending-k accumulation -> equal bit for bit (fp32 output), equal after the output
    rounding (bf16 output); and against a float64 product of the bf16-rounded operands."""
    from dispu_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M + K + N)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(K, N, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    st = _lib.stream_ptr(dev)
    ref = torch.empty(M, N, device=dev)
    _lib.check(L.dispu_linear_bf16(1, M, K, N, x.data_ptr(), K, 0, w.data_ptr(), N, 0, 0, b.data_ptr(), act, ref.data_ptr(), N, 0, None, 0, 0, None, 0, 0, st), "ref")
    bt = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    _lib.check(L.dispu_bf16_pack(K, N, w.data_ptr(), N, 1, bt.data_ptr(), st), "pack")
    assert torch.equal(bt, w.t().contiguous().to(torch.bfloat16))
    y = torch.empty(M, N, dtype=torch.bfloat16 if ybf else torch.float32, device=dev)
    _lib.check(L.dispu_linear_bf16_stream(M, K, N, x.data_ptr(), K, 0, bt.data_ptr(), K, b.data_ptr(), act, y.data_ptr(), N, ybf, 1, 0, st), "stream")
    if ybf:
        assert torch.equal(y, ref.to(torch.bfloat16))
    else:
        assert torch.equal(y, ref)
    exact = x.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double() + b.double()
    if act:
        exact = exact.clamp_min(0)
    bound = 2e-6 * (x.abs().double() @ w.abs().double()) + 1e-6
    tol = bound if not ybf else bound + exact.abs() * 2 ** -8
    assert bool(((y.double() - exact).abs() <= tol).all())
    # the straight (not transposed) pack, and shapes the streaming kernel refuses
    wt = torch.empty(K, N, dtype=torch.bfloat16, device=dev)
    _lib.check(L.dispu_bf16_pack(K, N, w.data_ptr(), N, 0, wt.data_ptr(), st), "pack0")
    assert torch.equal(wt, w.to(torch.bfloat16))
    assert L.dispu_linear_bf16_stream(M + 1, K, N, x.data_ptr(), K, 0, bt.data_ptr(), K, b.data_ptr(), act, y.data_ptr(), N, ybf, 1, 0, st) != 0
    # X stored as bf16 (the training step's bf16 activation storage): no conversion in the kernel, same products
    xb = x.to(torch.bfloat16)
    y2 = torch.empty_like(y)
    _lib.check(L.dispu_linear_bf16_stream(M, K, N, xb.data_ptr(), K, 1, bt.data_ptr(), K, b.data_ptr(), act, y2.data_ptr(), N, ybf, 1, 0, st), "stream bf16 x")
    assert torch.equal(y2, y)
    # k splits: fp32 partial products, added in order by dispu_linear_splitk_finish (reassociated: compared within the bound)
    if K % 64 == 0 and not ybf:
        nsp = 2 if K % 128 else 4
        parts = torch.empty(nsp, M, N, device=dev)
        _lib.check(L.dispu_linear_bf16_stream(M, K, N, x.data_ptr(), K, 0, bt.data_ptr(), K, None, 0, parts.data_ptr(), N, 0, nsp, M * N, st), "stream split")
        y3 = torch.empty(M, N, device=dev)
        _lib.check(L.dispu_linear_splitk_finish(M, N, nsp, parts.data_ptr(), M * N, b.data_ptr(), act, y3.data_ptr(), N, st), "finish")
        assert bool(((y3.double() - exact).abs() <= bound).all())


@pytest.mark.gpu
def test_bf16_step_with_and_without_the_streaming_kernel(steps, dev):
    """Trainer(dtype="bf16") routes its large dense products (after_conv forward / dX, the pair tensors' conv1) through
    dispu_linear_bf16_stream; Trainer.bf16_stream = False keeps them on dispu_linear_bf16.  Same operand roundings and k order -- the
    only difference is the k-split of after_conv's forward at few rows (fp32 partial sums added in order): coarse clouds equal to 1e-5,
    fine ones to 1e-4, gradients to 2e-3 of their norm (float atomics in the scatter gradients are part of that)."""
    from dispu_amd.train import Trainer
    tx, tg, r, P = steps["data"]
    res = {}
    for on in (True, False):
        tr = Trainer(params=P, device=dev, dtype="bf16")
        tr.bf16_stream = on
        tr.zero_grad()
        c, f = tr.forward(tx)
        tr.loss_backward(tg, r)
        tr.backward()
        torch.cuda.synchronize()
        res[on] = (N_(c).copy(), N_(f).copy(), N_(tr.flat_g).astype(np.float64))
    assert np.abs(res[True][0] - res[False][0]).max() <= 1e-5 and np.abs(res[True][1] - res[False][1]).max() <= 1e-4
    ga, gb = res[True][2], res[False][2]
    assert np.linalg.norm(ga - gb) <= 2e-3 * np.linalg.norm(gb), float(np.linalg.norm(ga - gb) / np.linalg.norm(gb))


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,st,acc", [(8192, 2048, 256, 0, 1), (4096, 128, 128, 3, 0), (131072, 128, 128, 3, 1), (1024, 256, 384, 0, 0), (8192, 256, 2048, 0, 1)])
def test_streaming_bf16_weight_gradient(M, K, N, st, acc):
    """dispu_linear_tn_bf16_stream: dW = X^T . Z with fp32- or bf16-stored operands against a float64 product of the bf16-rounded
    operands (the bound of test_linear_tn_bf16), accumulation into out, the bias gradient from the un-rounded Z."""
    from dispu_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M + K + N)
    x = torch.randn(M, K, device=dev, generator=g)
    z = torch.randn(M, N, device=dev, generator=g) * 0.1
    xs, zs = (x.to(torch.bfloat16), z.to(torch.bfloat16)) if st == 3 else (x, z)
    out0 = torch.randn(K, N, device=dev, generator=g)
    db0 = torch.randn(N, device=dev, generator=g)
    out, db = out0.clone(), db0.clone()
    need = L.dispu_linear_tn_bf16_stream_scratch_floats(M, K, N)
    assert need > 0
    sc = torch.empty(need, device=dev)
    _lib.check(L.dispu_linear_tn_bf16_stream(M, K, N, xs.data_ptr(), K, zs.data_ptr(), N, st, out.data_ptr(), N, acc, db.data_ptr(), sc.data_ptr(), need,
                                             _lib.stream_ptr(dev)), "tn stream")
    xr, zr = x.to(torch.bfloat16).double(), z.to(torch.bfloat16).double()
    want = xr.t() @ zr + (out0.double() if acc else 0)
    bound = 4e-6 * (xr.abs().t() @ zr.abs()) + 1e-6 * (1 + want.abs())
    assert bool(((out.double() - want).abs() <= bound).all()), float((out.double() - want).abs().max())
    zsum = (zs.double() if st == 3 else z.double()).sum(0) + (db0.double() if acc else 0)
    assert bool(((db.double() - zsum).abs() <= 1e-5 * (1 + z.abs().double().sum(0))).all())
    assert L.dispu_linear_tn_bf16_stream_scratch_floats(M, K + 1, N) == 0

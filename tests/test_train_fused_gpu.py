"""GPU parity of the round-3 training kernels (csrc/train_fused.hip, the masked GEMM epilogues, the stashing head chains) against
float64 torch autograd of the forward op each one differentiates -- 1e-5, like tests/test_train_gpu.py.  The reference has no
native code here (TF1 autodiff, DisPU/model.py:158-178); the forward ops are Common/ops.py:1012-1087 (PointShuffle2)."""
import ctypes as C
import ctypes

import numpy as np
import pytest
import torch

from oracle import train_oracle as T

pytestmark = pytest.mark.gpu
F64 = torch.float64
_KEEP = []


def N_(t):
    return t.detach().cpu().numpy()


def dv(a, dev, dtype=torch.float32):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dtype)
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release():
    yield
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    del _KEEP[:]


def p(t, off=0):
    return C.c_void_p(t.data_ptr() + 4 * off) if t is not None else C.c_void_p(0)


def close(a, ref, rel, what=""):
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(np.asarray(a, np.float64) - ref).max()
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e (rel %.2e)" % (what, err, scale, err / scale)


@pytest.fixture(scope="module")
def L():
    from dispu_amd import _lib
    return _lib


@pytest.mark.parametrize("M,Kc,Nout,mcols,acc", [(1000, 24, 168, 24, 1), (4096, 256, 64, 64, 0), (3000, 3, 64, 64, 0), (2048, 256, 2048, 0, 0),
                                                 (5000, 128, 128, 128, 1), (777, 64, 256, 100, 1), (8192, 256, 128, 128, 1)])
@pytest.mark.parametrize("kind", ["f32", "bf16"])
def test_linear_masked(dev, L, M, Kc, Nout, mcols, acc, kind):
    """dX[M, Nout] = mask(R1 + dZ[M, Kc] . W[Nout, Kc]^T): the dX product of a layer with the ReLU gradient of the layer below in
    its epilogue (tiled fp32 kernel, skinny kernel, bf16 kernel; masks over all / a prefix / none of the columns)."""
    rng = np.random.default_rng(M + Kc)
    dZ = rng.standard_normal((M, Kc)).astype(np.float32)
    W = rng.standard_normal((Nout, Kc)).astype(np.float32)
    R = rng.standard_normal((M, Nout)).astype(np.float32)
    Mk = rng.standard_normal((M, Nout)).astype(np.float32)
    Mk[rng.random((M, Nout)) < 0.2] = 0.0
    out = dv(R if acc else np.zeros_like(R), dev)
    fn = L.lib().dispu_linear_masked if kind == "f32" else L.lib().dispu_linear_bf16_masked
    tz, tw, tm = dv(dZ, dev), dv(W, dev), dv(Mk, dev)
    L.check(fn(1, M, Kc, Nout, p(tz), Kc, 0, p(tw), Kc, 0, 1, None, 0, p(out), Nout, 0, p(out) if acc else None, Nout if acc else 0, 0,
               p(tm) if mcols else None, Nout, mcols, L.stream_ptr(dev)), "linear_masked")
    if kind == "bf16":
        r16 = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float64).numpy()
        ref = r16(dZ) @ r16(W).T
    else:
        ref = dZ.astype(np.float64) @ W.astype(np.float64).T
    if acc:
        ref = ref + R
    keep = np.ones((M, Nout), bool)
    keep[:, :mcols] = Mk[:, :mcols] > 0
    ref = np.where(keep, ref, 0.0)
    got = N_(out)
    assert not got[~keep].any()                                    # masked entries are exactly zero
    close(got, ref, 2e-6 * max(Kc, 8) ** 0.5, "masked dX")


def test_mask3(dev, L):
    rng = np.random.default_rng(1)
    rows, n = 3000, 256
    d = rng.standard_normal((rows, n)).astype(np.float32)
    ys = [np.maximum(rng.standard_normal((rows, n)), 0).astype(np.float32) for _ in range(3)]
    outs = [torch.empty((rows, n), dtype=torch.float32, device=dev) for _ in range(3)]
    ty = [dv(y, dev) for y in ys]
    L.check(L.lib().dispu_mask3(rows, n, p(dv(d, dev)), n, p(ty[0]), n, p(ty[1]), n, p(ty[2]), n, p(outs[0]), p(outs[1]), p(outs[2]), n,
                                L.stream_ptr(dev)), "mask3")
    for o, y in zip(outs, ys):
        assert np.array_equal(N_(o), np.where(y > 0, d, 0).astype(np.float32))


def _cloud(rng, B, n, k):
    xyz = rng.standard_normal((B, n, 3)).astype(np.float32) * 0.3
    d = ((xyz[:, :, None, :] - xyz[:, None, :, :]) ** 2).sum(-1)
    idx = np.argsort(d, axis=-1, kind="stable")[:, :, :k].astype(np.int32)
    return xyz, idx


def test_wnet_bn_stats_and_grad(dev, L):
    """weight_net_hidden in training mode (ops.py:181-191) without its stored input: batch statistics / folded scale+shift / moving
    statistics against float64, then every gradient (dWw, dbw ~ 0, dgamma, dbeta, dxyz of both points of a pair)."""
    rng = np.random.default_rng(3)
    B, n, k, t = 2, 256, 16, 16
    xyz, idx = _cloud(rng, B, n, k)
    Ww = (rng.standard_normal((3, t)) * 2).astype(np.float32)
    bw = (rng.standard_normal(t) * 0.1).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, t).astype(np.float32)
    beta = (rng.standard_normal(t) * 0.3).astype(np.float32)
    mm0, mv0 = rng.standard_normal(t).astype(np.float32), rng.uniform(0.5, 1.5, t).astype(np.float32)
    rows = B * n
    lib, st = L.lib(), L.stream_ptr(dev)
    di, dx = dv(idx, dev, torch.int32), dv(xyz, dev)
    tw, tb, tg, tbe = dv(Ww, dev), dv(bw, dev), dv(gamma, dev), dv(beta, dev)
    mm, mv = dv(mm0, dev), dv(mv0, dev)
    stats = torch.empty(48, dtype=torch.float32, device=dev)
    scale, shift = torch.empty(16, dtype=torch.float32, device=dev), torch.empty(16, dtype=torch.float32, device=dev)
    nb = lib.dispu_ps_wnet_scratch_bytes(rows)
    sc = torch.empty(nb // 8 + 1, dtype=torch.float64, device=dev)
    L.check(lib.dispu_ps_wnet_bn_stats(rows, n, k, t, p(di), p(dx), p(tw), p(tb), p(tg), p(tbe), 1e-3, 0.95, p(stats), p(scale), p(shift),
                                       p(mm), p(mv), p(sc), nb, st), "wnet_bn_stats")
    # float64 reference through the oracle's batch_norm
    xt = torch.tensor(xyz, dtype=F64, requires_grad=True)
    it = torch.tensor(idx.astype(np.int64))
    Pt = {"s/gamma": torch.tensor(gamma, dtype=F64, requires_grad=True), "s/beta": torch.tensor(beta, dtype=F64, requires_grad=True),
          "s/moving_mean": torch.tensor(mm0, dtype=F64), "s/moving_variance": torch.tensor(mv0, dtype=F64)}
    wt = torch.tensor(Ww, dtype=F64, requires_grad=True)
    bt = torch.tensor(bw, dtype=F64, requires_grad=True)
    off = T.gather(xt, it) - xt[:, :, None, :]
    wl = off @ wt + bt
    state = {}
    wv = torch.relu(T.batch_norm(Pt, "s/", wl, True, state))
    flat = wl.reshape(-1, t).detach()
    close(N_(stats)[:16], flat.mean(0).numpy(), 1e-5, "mean")
    close(N_(stats)[16:32], flat.var(0, unbiased=False).numpy(), 1e-5, "var")
    close(N_(mm), state["moving_mean"].numpy(), 1e-6, "moving_mean")
    close(N_(mv), state["moving_variance"].numpy(), 1e-6, "moving_variance")
    # the folded form is what dispu_ps_weight_net applies: wv from the device == oracle forward
    wvd = torch.empty((rows * k, t), dtype=torch.float32, device=dev)
    L.check(lib.dispu_ps_weight_net(rows, n, k, t, p(di), p(dx), p(tw), p(tb), p(scale), p(shift), p(wvd), st), "weight_net")
    close(N_(wvd).reshape(B, n, k, t), wv.detach().numpy(), 1e-5, "wv (batch statistics)")
    g = rng.standard_normal((B, n, k, t)).astype(np.float32)
    wv.backward(torch.tensor(g, dtype=F64))
    dWw, dbw = torch.zeros((3, t), dtype=torch.float32, device=dev), torch.zeros(t, dtype=torch.float32, device=dev)
    dga, dbe = torch.zeros(t, dtype=torch.float32, device=dev), torch.zeros(t, dtype=torch.float32, device=dev)
    dxyz = torch.zeros((rows, 3), dtype=torch.float32, device=dev)
    sums = torch.empty(32, dtype=torch.float32, device=dev)
    L.check(lib.dispu_ps_wnet_grad(rows, n, k, t, p(di), p(dx), p(tw), p(tb), p(stats), p(scale), p(shift), p(tg), p(dv(g.reshape(-1, t), dev)),
                                   p(dWw), p(dbw), p(dga), p(dbe), p(dxyz), p(sums), p(sc), nb, st), "wnet_grad")
    close(N_(dWw), wt.grad.numpy(), 2e-5, "dWw")
    close(N_(dga), Pt["s/gamma"].grad.numpy(), 1e-5, "dgamma")
    close(N_(dbe), Pt["s/beta"].grad.numpy(), 1e-5, "dbeta")
    close(N_(dxyz).reshape(B, n, 3), xt.grad.numpy(), 2e-5, "dxyz")
    assert np.abs(N_(dbw)).max() <= 1e-4 * np.abs(wt.grad.numpy()).max()      # a bias in front of BatchNorm: identically 0


@pytest.mark.parametrize("B,n,k", [(2, 256, 16), (3, 1024, 16), (1, 100, 7), (2, 4096, 4)])
def test_knn_invert(dev, L, B, n, k):
    rng = np.random.default_rng(n + k)
    idx = rng.integers(0, n, (B, n, k)).astype(np.int32)
    idx[0, :, 0] = 5 % n                                                       # one heavily shared neighbour, many empty lists
    off = torch.empty((B, n + 1), dtype=torch.int32, device=dev)
    inv = torch.empty((B, n * k), dtype=torch.int32, device=dev)
    L.check(L.lib().dispu_knn_invert(B, n, k, p(dv(idx, dev, torch.int32)), p(off), p(inv), L.stream_ptr(dev)), "knn_invert")
    o, iv = N_(off), N_(inv)
    for b in range(B):
        flat = idx[b].reshape(-1)
        order = np.argsort(flat, kind="stable")                                # pair ids grouped by target, ascending inside a group
        assert np.array_equal(iv[b], order.astype(np.int32))
        assert np.array_equal(o[b], np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=n))]).astype(np.int32))


def test_conv0_per_source_point_backward(dev, L):
    """h0 = relu(G[j] - A[i]) with G = feat.Wf + xyz.(Wc + Wr) + b0, A = xyz.Wc (csrc/mlp_misc.hip:ps_prep): dz0 -> dG / dAneg through the
    inverted graph (no atomics: run twice, bit-identical), then dxyz and dW0[0:6] (ps_prep_grad) against float64 autograd."""
    rng = np.random.default_rng(11)
    B, n, k, c = 2, 256, 16, 128
    xyz, idx = _cloud(rng, B, n, k)
    rows = B * n
    W0 = (rng.standard_normal((134, c)) * 0.2).astype(np.float32)
    dz0 = rng.standard_normal((rows * k, c)).astype(np.float32)
    dz0[rng.random((rows * k, c)) < 0.4] = 0.0
    lib, st = L.lib(), L.stream_ptr(dev)
    di = dv(idx, dev, torch.int32)
    off = torch.empty((B, n + 1), dtype=torch.int32, device=dev)
    inv = torch.empty((B, n * k), dtype=torch.int32, device=dev)
    L.check(lib.dispu_knn_invert(B, n, k, p(di), p(off), p(inv), st), "knn_invert")
    tz = dv(dz0, dev)
    res = []
    for _ in range(2):
        dG, dA = torch.empty((rows, c), dtype=torch.float32, device=dev), torch.empty((rows, c), dtype=torch.float32, device=dev)
        L.check(lib.dispu_ps_conv0_gather_grad(rows, n, k, c, p(di), p(off), p(inv), p(tz), c, None, 0, None, 0, p(dG), c, p(dA), c, st),
                "conv0_gather_grad")
        res.append((N_(dG).copy(), N_(dA).copy()))
        _KEEP.extend([dG, dA])
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    z = dz0.reshape(B, n, k, c).astype(np.float64)
    refG = np.zeros((B, n, c))
    for b in range(B):
        np.add.at(refG[b], idx[b].reshape(-1), z[b].reshape(-1, c))
    close(res[0][0].reshape(B, n, c), refG, 1e-5, "dG")
    close(res[0][1].reshape(B, n, c), -z.sum(2), 1e-5, "dAneg")
    # with the ReLU decision of h0 = relu(G[j] - A[i]) re-derived inside the kernel
    Gm, Am = rng.standard_normal((rows, c)).astype(np.float32), rng.standard_normal((rows, c)).astype(np.float32)
    dGm, dAm = torch.empty((rows, c), dtype=torch.float32, device=dev), torch.empty((rows, c), dtype=torch.float32, device=dev)
    L.check(lib.dispu_ps_conv0_gather_grad(rows, n, k, c, p(di), p(off), p(inv), p(tz), c, p(dv(Gm, dev)), c, p(dv(Am, dev)), c, p(dGm), c, p(dAm), c, st),
            "conv0_gather_grad(masked)")
    gj = np.stack([Gm.reshape(B, n, c)[b][idx[b]] for b in range(B)])                     # [B, n, k, c]
    keep = (gj - Am.reshape(B, n, 1, c)) > 0
    zm = z * keep
    refGm = np.zeros((B, n, c))
    for b in range(B):
        np.add.at(refGm[b], idx[b].reshape(-1), zm[b].reshape(-1, c))
    close(N_(dGm).reshape(B, n, c), refGm, 1e-5, "dG (masked)")
    close(N_(dAm).reshape(B, n, c), -zm.sum(2), 1e-5, "dAneg (masked)")
    # xyz side: autograd of sum(dz0 * (G[j] - A[i])) w.r.t. xyz and W0[0:6], feature term excluded
    xt = torch.tensor(xyz, dtype=F64, requires_grad=True)
    wt = torch.tensor(W0[:6], dtype=F64, requires_grad=True)
    it = torch.tensor(idx.astype(np.int64))
    Gx = xt @ (wt[0:3] + wt[3:6])
    Ax = xt @ wt[0:3]
    pre = T.gather(Gx, it) - Ax[:, :, None, :]
    (pre * torch.tensor(z)).sum().backward()
    dxyz = torch.zeros((rows, 3), dtype=torch.float32, device=dev)
    dW0 = torch.zeros((134, c), dtype=torch.float32, device=dev)
    L.check(lib.dispu_ps_prep_grad(rows, c, p(dv(xyz, dev)), p(dv(W0, dev)), p(dv(res[0][0], dev)), c, p(dv(res[0][1], dev)), c, p(dxyz), p(dW0), st),
            "ps_prep_grad")
    close(N_(dxyz).reshape(B, n, 3), xt.grad.numpy(), 2e-5, "dxyz")
    close(N_(dW0)[:6], wt.grad.numpy(), 2e-5, "dW0[0:6]")
    assert not N_(dW0)[6:].any()


def test_skip_max_grad(dev, L):
    """max over the 16 neighbours of [xyz_j - xyz_i | xyz_j | feat_j] (ops.py:1049) backward without the grouped tensor; duplicate
    neighbours make exact ties, which share the gradient evenly (math_grad._MinOrMaxGrad == oracle max_even)."""
    rng = np.random.default_rng(13)
    B, n, k, cf = 2, 256, 16, 128
    xyz, idx = _cloud(rng, B, n, k)
    idx[:, ::3, 5] = idx[:, ::3, 2]                                           # repeated neighbour: every channel it wins is a 2-way tie
    feat = rng.standard_normal((B, n, cf)).astype(np.float32)
    rows = B * n
    lib, st = L.lib(), L.stream_ptr(dev)
    di, dx, df = dv(idx, dev, torch.int32), dv(xyz, dev), dv(feat, dev)
    gmax = torch.zeros((rows, 144), dtype=torch.float32, device=dev)
    L.check(lib.dispu_ps_skip_max(rows, n, k, cf, p(di), p(dx), p(df), cf, p(gmax), 144, st), "skip_max")
    xt = torch.tensor(xyz, dtype=F64, requires_grad=True)
    ft = torch.tensor(feat, dtype=F64, requires_grad=True)
    it = torch.tensor(idx.astype(np.int64))
    gx = T.gather(xt, it)
    grouped = torch.cat([gx - xt[:, :, None, :], gx, T.gather(ft, it)], -1)
    ref = T.max_even(grouped, 2)
    close(N_(gmax)[:, :134].reshape(B, n, 134), ref.detach().numpy(), 1e-6, "gmax")
    g = rng.standard_normal((rows, 136)).astype(np.float32)
    ref.backward(torch.tensor(g[:, :134].reshape(B, n, 134), dtype=F64))
    dxyz = torch.zeros((rows, 3), dtype=torch.float32, device=dev)
    dfeat = torch.zeros((rows, cf), dtype=torch.float32, device=dev)
    L.check(lib.dispu_ps_skip_max_grad(rows, n, k, cf, p(di), p(dx), p(df), cf, p(gmax), 144, p(dv(g, dev)), 136, p(dxyz), p(dfeat), cf, 0, st),
            "skip_max_grad")
    close(N_(dfeat).reshape(B, n, cf), ft.grad.numpy(), 1e-5, "dfeat")
    close(N_(dxyz).reshape(B, n, 3), xt.grad.numpy(), 1e-5, "dxyz")
    # feat_is_relu: ReLU features (half of them zeros, so many maxima are 16-way ties at 0); equal to the full gradient after relu_grad
    featr = np.maximum(feat, 0).astype(np.float32)
    featr[:, :, :8] = 0.0
    dfr = dv(featr, dev)
    L.check(lib.dispu_ps_skip_max(rows, n, k, cf, p(di), p(dx), p(dfr), cf, p(gmax), 144, st), "skip_max")
    outs = []
    for flag in (0, 1):
        dxyz2 = torch.zeros((rows, 3), dtype=torch.float32, device=dev)
        dfeat2 = torch.zeros((rows, cf), dtype=torch.float32, device=dev)
        L.check(lib.dispu_ps_skip_max_grad(rows, n, k, cf, p(di), p(dx), p(dfr), cf, p(gmax), 144, p(dv(g, dev)), 136, p(dxyz2), p(dfeat2), cf, flag, st),
                "skip_max_grad")
        outs.append((N_(dfeat2) * (featr.reshape(rows, cf) > 0), N_(dxyz2)))
    close(outs[1][0], outs[0][0].astype(np.float64), 1e-5, "dfeat after relu_grad")
    close(outs[1][1], outs[0][1].astype(np.float64), 1e-5, "dxyz")


def test_point_matmul_grad_relu(dev, L):
    rng = np.random.default_rng(7)
    rows, k, c, t = 37, 16, 128, 16
    X2 = np.maximum(rng.standard_normal((rows * k, c)), 0).astype(np.float32)        # h1 = a ReLU output
    wv = rng.standard_normal((rows * k, t)).astype(np.float32)
    do = rng.standard_normal((rows, c * t)).astype(np.float32)
    dX2 = torch.empty((rows * k, c), dtype=torch.float32, device=dev)
    dwv = torch.empty((rows * k, t), dtype=torch.float32, device=dev)
    L.check(L.lib().dispu_ps_point_matmul_grad_relu(rows, k, c, t, p(dv(X2, dev)), c, p(dv(wv, dev)), p(dv(do, dev)), c * t, p(dX2), c,
                                                    p(dwv), L.stream_ptr(dev)), "point_matmul_grad_relu")
    xt = torch.tensor(X2.reshape(rows, k, c), dtype=F64, requires_grad=True)
    wt = torch.tensor(wv.reshape(rows, k, t), dtype=F64, requires_grad=True)
    (xt.transpose(1, 2) @ wt).reshape(rows, c * t).backward(torch.tensor(do, dtype=F64))
    close(N_(dX2).reshape(rows, k, c), xt.grad.numpy() * (X2.reshape(rows, k, c) > 0), 1e-5, "dz1")
    close(N_(dwv).reshape(rows, k, t), wt.grad.numpy(), 1e-5, "dwv")


@pytest.mark.parametrize("rows", [384, 96, 12288, 24576])       # 32-row workgroups (< 192 of 64 rows; 96: only 32 divides), 64-row, 128-row
@pytest.mark.parametrize("shape,mode", [((256, 128, 256, 64), 0), ((256, 256, 256, 64), 1)])
def test_mlp_chain_stash_equals_separate_launches(dev, L, shape, mode, rows):
    """dispu_mlp_chain_stash: the stashed Y1 / Y2 / Y3 / Z and the head output are bit-identical to the dispu_linear /
    dispu_linear_small_n launches of the same layers (the training forward of the two head chains, ops.py:1089-1110, 1186-1192)."""
    rng = np.random.default_rng(sum(shape))
    K0, N1, N2, N3 = shape
    X = rng.standard_normal((rows, K0)).astype(np.float32)
    Ws = [(rng.standard_normal((a, b)) / np.sqrt(a)).astype(np.float32) for a, b in ((K0, N1), (N1, N2), (N2, N3), (N3, 3))]
    bs = [(rng.standard_normal(b) * 0.1).astype(np.float32) for b in (N1, N2, N3, 3)]
    R = rng.standard_normal((rows, 3)).astype(np.float32)
    lib, st = L.lib(), L.stream_ptr(dev)
    tx, tr = dv(X, dev), dv(R, dev)
    tw, tb = [dv(w, dev) for w in Ws], [dv(b, dev) for b in bs]
    E = lambda n: torch.empty((rows, n), dtype=torch.float32, device=dev)
    y1, y2, y3, z, out = E(N1), E(N2), E(N3), E(3), E(3)
    L.check(lib.dispu_mlp_chain_stash(rows, K0, N1, N2, N3, p(tx), K0, p(tw[0]), p(tb[0]), p(tw[1]), p(tb[1]), p(tw[2]), p(tb[2]), p(tw[3]),
                                      p(tb[3]), p(y1), N1, p(y2), N2, p(y3), N3, p(z), 3, mode, p(tr) if mode else None, 3, p(out), 3, st),
            "mlp_chain_stash")
    r1, r2, r3, rz, ro = E(N1), E(N2), E(N3), E(3), E(3)
    lin = lambda x, k, w, b, act, y, n: L.check(lib.dispu_linear(1, rows, k, n, p(x), k, 0, p(w), n, 0, 0, p(b), act, p(y), n, 0, None, 0, 0,
                                                                 None, 0, 0, st), "linear")
    lin(tx, K0, tw[0], tb[0], 1, r1, N1)
    lin(r1, N1, tw[1], tb[1], 1, r2, N2)
    lin(r2, N2, tw[2], tb[2], 1, r3, N3)
    L.check(lib.dispu_linear_small_n(rows, N3, 3, p(r3), N3, p(tw[3]), p(tb[3]), 0, None, 0, p(rz), 3, st), "head z")
    L.check(lib.dispu_linear_small_n(rows, N3, 3, p(r3), N3, p(tw[3]), p(tb[3]), mode, p(tr) if mode else None, 3, p(ro), 3, st), "head")
    for a, b, name in ((y1, r1, "Y1"), (y2, r2, "Y2"), (y3, r3, "Y3"), (z, rz, "Z"), (out, ro, "out")):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("shape,fine,rows", [((256, 128, 256), False, 384), ((256, 256, 256), True, 384), ((256, 128, 256), False, 8192),
                                             ((256, 256, 256), True, 64),
                                             # 64-row workgroups (>= 192 of them) and a row count only 32-row workgroups tile
                                             ((256, 256, 256), True, 16384), ((256, 128, 256), False, 12288), ((256, 128, 256), False, 96)])
def test_mlp_chain_grad(dev, L, shape, fine, rows):
    """dispu_mlp_chain_grad (csrc/mlp_chain_bwd.hip): the four dX products of a head chain's backward in one launch against float64
    autograd of the chain X -> relu(X W1) -> relu(. W2) -> relu(. W3) -> . W4 (ops.py:1186-1192, 1089-1108, 1079-1083).  coarse: the
    chain input is a ReLU output (one mask) and dY1 accumulates onto an existing gradient (R aliases D1); fine: the chain input is the
    sum of three ReLU outputs, so dX leaves through three masks (dispu_mask3 folded in)."""
    rng = np.random.default_rng(rows + sum(shape) + int(fine))
    K0, N1, N2 = shape
    Ws = [(rng.standard_normal((a, b)) / np.sqrt(a)).astype(np.float32) for a, b in ((K0, N1), (N1, N2), (N2, 64), (64, 3))]
    bs = [(rng.standard_normal(b) * 0.1).astype(np.float32) for b in (N1, N2, 64)]
    if fine:
        parts = [np.maximum(rng.standard_normal((rows, K0)), 0).astype(np.float32) for _ in range(3)]
        X = (parts[0] + parts[1] + parts[2]).astype(np.float32)
    else:
        parts = [np.maximum(rng.standard_normal((rows, K0)), 0).astype(np.float32)]
        X = parts[0]
    dZ = rng.standard_normal((rows, 3)).astype(np.float32)
    Racc = rng.standard_normal((rows, N1)).astype(np.float32) if not fine else None
    Racc2 = rng.standard_normal((rows, N1)).astype(np.float32) if (not fine and rows != 8192) else None
    # float64 reference: activations as the fp32 forward stashes them (the masks), gradients by autograd
    xt = torch.tensor(X, dtype=F64, requires_grad=True)
    y1 = torch.relu(xt @ torch.tensor(Ws[0], dtype=F64) + torch.tensor(bs[0], dtype=F64)); y1.retain_grad()
    y2 = torch.relu(y1 @ torch.tensor(Ws[1], dtype=F64) + torch.tensor(bs[1], dtype=F64)); y2.retain_grad()
    y3 = torch.relu(y2 @ torch.tensor(Ws[2], dtype=F64) + torch.tensor(bs[2], dtype=F64)); y3.retain_grad()
    z = y3 @ torch.tensor(Ws[3], dtype=F64)
    extra = (y1 * torch.tensor(Racc, dtype=F64)).sum() if Racc is not None else 0.0        # other consumers of Y1: gradients R, R2 arrive at Y1
    if Racc2 is not None:
        extra = extra + (y1 * torch.tensor(Racc2, dtype=F64)).sum()
    ((z * torch.tensor(dZ, dtype=F64)).sum() + extra).backward()
    Y1, Y2, Y3 = (N_(t).astype(np.float32) for t in (y1, y2, y3))
    # a unit whose float64 pre-activation is within rounding of 0 may be masked differently in fp32: keep the masks float64-exact
    lib, st = L.lib(), L.stream_ptr(dev)
    tY1, tY2, tY3, tdz = dv(Y1, dev), dv(Y2, dev), dv(Y3, dev), dv(dZ, dev)
    tW4 = dv(Ws[3], dev)
    tWt = [dv(np.ascontiguousarray(w.T), dev) for w in Ws[:3]]
    E = lambda n: torch.full((rows, n), 7.0, dtype=torch.float32, device=dev)
    d3, d2 = E(64), E(N2)
    d1 = dv(Racc, dev) if Racc is not None else E(N1)
    masks = [dv(m, dev) for m in parts]
    outs = [E(K0) for _ in parts]
    pm = lambda i: p(masks[i]) if i < len(masks) else None
    po = lambda i: p(outs[i]) if i < len(outs) else None
    L.check(lib.dispu_mlp_chain_grad(rows, K0, N1, N2, p(tdz), 3, p(tW4), p(tWt[2]), p(tWt[1]), p(tWt[0]), p(tY3), 64, p(tY2), N2, p(tY1), N1,
                                     p(d1) if Racc is not None else None, N1, p(dv(Racc2, dev)) if Racc2 is not None else None, N1, p(d3), 64, p(d2), N2, p(d1), N1, pm(0), pm(1), pm(2), K0,
                                     po(0), po(1), po(2), K0, st), "mlp_chain_grad")
    # autograd's .grad of a ReLU OUTPUT is the gradient before that ReLU's own mask; the kernel's D is after it
    close(N_(d3), y3.grad.numpy() * (Y3 > 0), 1e-5, "dY3")
    close(N_(d2), y2.grad.numpy() * (Y2 > 0), 1e-5, "dY2")
    close(N_(d1), y1.grad.numpy() * (Y1 > 0), 1e-5, "dY1")
    for m, o, name in zip(parts, outs, "abc"):
        close(N_(o), xt.grad.numpy() * (m > 0), 1e-5, "dX through mask " + name)


@pytest.mark.parametrize("C,B,n", [(24, 2, 256), (48, 2, 256), (48, 1, 100), (24, 3, 36)])
def test_edge_dense_conv_grad(dev, L, C, B, n):
    """dense_conv + get_edge_feature backward in one launch (csrc/edge_bwd.hip: forward recomputed on chip) against float64 autograd
    of oracle/train_oracle.py:dense_conv; duplicate neighbours make exact arg-max ties (shared evenly); n * B not a multiple of the
    8-point tile exercises the ragged last tile."""
    rng = np.random.default_rng(C + n)
    k = 16
    F = rng.standard_normal((B, n, C)).astype(np.float32)
    idx = rng.integers(0, n, (B, n, k + 1)).astype(np.int32)                  # column 0 = "self" (dropped: ioff = 1), as knn_feat returns
    idx[:, ::5, 7] = idx[:, ::5, 3]
    scope = "s"
    shapes = {"/l0": (2 * C, 24), "/l1": (24 + C, 24), "/l2": (48 + C, 24)}
    P = {}
    for name, (a, b) in shapes.items():
        P[scope + name + "/weights"] = (rng.standard_normal((a, b)) / np.sqrt(a)).astype(np.float32)
        P[scope + name + "/biases"] = (rng.standard_normal(b) * 0.1).astype(np.float32)
    Pt = {kk: torch.tensor(v, dtype=F64, requires_grad=True) for kk, v in P.items()}
    ft = torch.tensor(F, dtype=F64, requires_grad=True)
    out = T.dense_conv(Pt, scope, ft, torch.tensor(idx[:, :, 1:].astype(np.int64)))
    rows = B * n
    lib, st = L.lib(), L.stream_ptr(dev)
    tF, ti = dv(F.reshape(rows, C), dev), dv(idx.reshape(rows, k + 1), dev, torch.int32)
    tp = {kk: dv(v, dev) for kk, v in P.items()}
    # forward of the device kernel == oracle forward (same arithmetic as the recompute inside the backward kernel)
    Y = torch.empty((rows, 72 + C), dtype=torch.float32, device=dev)
    L.check(lib.dispu_edge_dense_conv(rows, n, C, p(tF), C, p(ti), k + 1, 1, p(tp["s/l0/weights"]), p(tp["s/l0/biases"]), p(tp["s/l1/weights"]),
                                      p(tp["s/l1/biases"]), p(tp["s/l2/weights"]), p(tp["s/l2/biases"]), p(Y), 72 + C, st), "edge_dense_conv")
    close(N_(Y).reshape(B, n, 72 + C), out.detach().numpy(), 1e-5, "forward")
    g = rng.standard_normal((rows, 72 + C)).astype(np.float32)
    out.backward(torch.tensor(g.reshape(B, n, 72 + C), dtype=F64))
    dF = torch.zeros((rows, C), dtype=torch.float32, device=dev)
    dW = {kk: torch.zeros(v.shape, dtype=torch.float32, device=dev) for kk, v in P.items()}
    need = lib.dispu_edge_dense_conv_grad_scratch_floats(rows, C)
    sc = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
    for rep in range(2):                                                      # accumulating entry: the second call doubles everything
        L.check(lib.dispu_edge_dense_conv_grad(rows, n, C, p(tF), C, p(ti), k + 1, 1, p(tp["s/l0/weights"]), p(tp["s/l0/biases"]),
                                               p(tp["s/l1/weights"]), p(tp["s/l1/biases"]), p(tp["s/l2/weights"]), p(tp["s/l2/biases"]),
                                               p(dv(g, dev)), 72 + C, p(dF), C, p(dW["s/l0/weights"]), p(dW["s/l0/biases"]), p(dW["s/l1/weights"]),
                                               p(dW["s/l1/biases"]), p(dW["s/l2/weights"]), p(dW["s/l2/biases"]), p(sc), need, st),
                "edge_dense_conv_grad")
        close(N_(dF).reshape(B, n, C), (rep + 1) * ft.grad.numpy(), 2e-5, "dF")
        for kk in P:
            close(N_(dW[kk]), (rep + 1) * Pt[kk].grad.numpy(), 2e-5, kk)
    # the two-halves form (weight-gradient sums on another stream in the training step): the same weight gradients, bit for bit
    dF2 = torch.zeros((rows, C), dtype=torch.float32, device=dev)
    dW2 = {kk: torch.zeros(v.shape, dtype=torch.float32, device=dev) for kk, v in P.items()}
    side = torch.cuda.Stream(device=dev)
    for rep in range(2):
        L.check(lib.dispu_edge_dense_conv_grad_partials(rows, n, C, p(tF), C, p(ti), k + 1, 1, p(tp["s/l0/weights"]), p(tp["s/l0/biases"]),
                                                        p(tp["s/l1/weights"]), p(tp["s/l1/biases"]), p(tp["s/l2/weights"]), p(tp["s/l2/biases"]),
                                                        p(dv(g, dev)), 72 + C, p(dF2), C, p(sc), need, st), "edge_dense_conv_grad_partials")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        side.wait_event(ev)
        L.check(lib.dispu_edge_dense_conv_grad_reduce(rows, C, p(sc), need, p(dW2["s/l0/weights"]), p(dW2["s/l0/biases"]), p(dW2["s/l1/weights"]),
                                                      p(dW2["s/l1/biases"]), p(dW2["s/l2/weights"]), p(dW2["s/l2/biases"]),
                                                      ctypes.c_void_p(side.cuda_stream)), "edge_dense_conv_grad_reduce")
        torch.cuda.current_stream(dev).wait_stream(side)                      # the next pass overwrites the partials
    for kk in P:
        assert torch.equal(dW2[kk], dW[kk]), kk
    close(N_(dF2).reshape(B, n, C), 2 * ft.grad.numpy(), 2e-5, "dF (two halves)")


# ---- round 4: the non-local cell's attention without the [B, M, M] tensor (csrc/attention_train.hip) ----------------------------
@pytest.mark.parametrize("b,m,strided", [(8, 1024, True), (2, 256, False), (1, 96, True), (16, 1024, False), (32, 1024, True), (3, 32, False), (64, 512, False)])
def test_attention_train_forward_and_backward(dev, L, b, m, strided):
    """softmax(Q.K^T / 8).V of Common/ops.py:326-339 and its gradients w.r.t. Q, K, V against float64 autograd of the SAME
    three TF ops (matmul(transpose_b), softmax, matmul).  The backward recomputes the probabilities from Q, K and the forward's
    per-row log-sum-exp; b / m sweep the three workgroup shapes (1, 2, 4 MFMA waves) and the strided K|V layout of the trainer."""
    rng = np.random.default_rng(b * 1000 + m)
    q = rng.standard_normal((b, m, 64)).astype(np.float32) * 1.5
    kv = rng.standard_normal((b, m, 128)).astype(np.float32) * 1.5
    do = rng.standard_normal((b, m, 64)).astype(np.float32)
    tq = torch.tensor(q, dtype=F64, requires_grad=True)
    tkv = torch.tensor(kv, dtype=F64, requires_grad=True)
    att = torch.softmax(torch.matmul(tq, tkv[..., :64].transpose(1, 2)) / 8.0, dim=-1)
    out = torch.matmul(att, tkv[..., 64:])
    out.backward(torch.tensor(do, dtype=F64))
    st = L.stream_ptr(dev)
    Q, dO = dv(q, dev), dv(do, dev)
    if strided:
        KV = dv(kv, dev)
        Kp, Vp, ld = p(KV), p(KV, 64), 128
        dKV = torch.full((b, m, 128), float("nan"), device=dev)
        dKp, dVp = p(dKV), p(dKV, 64)
    else:
        Kt, Vt = dv(kv[..., :64].copy(), dev), dv(kv[..., 64:].copy(), dev)
        Kp, Vp, ld = p(Kt), p(Vt), 64
        dKt, dVt = torch.full((b, m, 64), float("nan"), device=dev), torch.full((b, m, 64), float("nan"), device=dev)
        dKp, dVp = p(dKt), p(dVt)
    O = torch.full((b, m, 64), float("nan"), device=dev)
    lse = torch.full((b, m), float("nan"), device=dev)
    dQ = torch.full((b, m, 64), float("nan"), device=dev)
    dvec = torch.empty((b, m), device=dev)
    L.check(L.lib().dispu_attention_fwd_lse(b, m, m, 64, p(Q), 64, Kp, ld, Vp, ld, 0.125, p(O), 64, p(lse), st), "attention_fwd_lse")
    L.check(L.lib().dispu_attention_bwd(b, m, m, 64, p(Q), 64, Kp, ld, Vp, ld, 0.125, p(O), 64, p(lse), p(dO), 64, p(dQ), 64,
                                        dKp, ld, dVp, ld, p(dvec), st), "attention_bwd")
    torch.cuda.synchronize()
    close(N_(O), out.detach().numpy(), 1e-5, "O")
    logits = (torch.matmul(tq, tkv[..., :64].transpose(1, 2)) / 8.0).detach()
    want_lse = (torch.logsumexp(logits, dim=-1) / np.log(2.0)).numpy()
    assert np.abs(N_(lse) - want_lse).max() <= 1e-5 * max(1.0, np.abs(want_lse).max())
    close(N_(dvec), (torch.tensor(do, dtype=F64) * out.detach()).sum(-1).numpy(), 1e-5, "D")
    close(N_(dQ), tq.grad.numpy(), 1e-5, "dQ")
    gkv = tkv.grad.numpy()
    if strided:
        close(N_(dKV)[..., :64], gkv[..., :64], 1e-5, "dK")
        close(N_(dKV)[..., 64:], gkv[..., 64:], 1e-5, "dV")
    else:
        close(N_(dKt), gkv[..., :64], 1e-5, "dK")
        close(N_(dVt), gkv[..., 64:], 1e-5, "dV")
    # deterministic: no float atomics anywhere in the two backward kernels
    dQ2 = torch.empty_like(dQ)
    L.check(L.lib().dispu_attention_bwd(b, m, m, 64, p(Q), 64, Kp, ld, Vp, ld, 0.125, p(O), 64, p(lse), p(dO), 64, p(dQ2), 64,
                                        dKp, ld, dVp, ld, p(dvec), st), "attention_bwd")
    torch.cuda.synchronize()
    assert torch.equal(dQ, dQ2)


def test_attention_train_refuses_unsupported_shapes(dev, L):
    t = torch.zeros((1, 64, 64), device=dev)
    v = torch.zeros(64, device=dev)
    st = L.stream_ptr(dev)
    bad = L.lib().dispu_attention_fwd_lse(1, 48, 64, 64, p(t), 64, p(t), 64, p(t), 64, 0.125, p(t), 64, p(v), st)     # m % 32
    assert bad != 0
    bad = L.lib().dispu_attention_fwd_lse(1, 64, 64, 32, p(t), 64, p(t), 64, p(t), 64, 0.125, p(t), 64, p(v), st)     # d != 64
    assert bad != 0
    bad = L.lib().dispu_attention_fwd_lse(1, 64, 64, 64, p(t), 66, p(t), 64, p(t), 64, 0.125, p(t), 64, p(v), st)     # unaligned rows
    assert bad != 0


# ---------------------------------------------------------------------------------------- round 6: grouped split reductions ----
def test_deferred_reductions_in_one_grouped_launch_equal_the_products_own(dev, L):
    """dispu_tn_defer + dispu_tn_reduce_grouped (csrc/train_gemm.hip): five weight-gradient products of the four TN kernels (tiled fp32,
    narrow fp32, bf16, streaming bf16 with fp32- and bf16-stored operands; accumulate on and off, with and without a bias gradient,
    a non-multiple-of-4 width) leave their reductions to ONE grouped launch -- bit-identical to each product reducing itself.  The sink is
    one-shot: the call after a deferred one reduces itself again; batched products ignore it."""
    lib = L.lib()
    st = L.stream_ptr(dev)
    rng = np.random.default_rng(66)
    jobs = [  # (kind, M, K, N, accumulate, bias)
        ("f32", 8192, 256, 128, 1, True), ("f32", 8192, 64, 24, 1, True), ("f32", 5000, 134, 255, 0, False),
        ("bf16", 8192, 256, 64, 1, True), ("stream", 8192, 2048, 256, 1, True), ("stream16", 16384, 128, 128, 0, True)]
    descs = (L.TnReduceDesc * len(jobs))()
    want, got, keep = [], [], []
    for i, (kind, M, K, N, acc, bias) in enumerate(jobs):
        x = dv(rng.standard_normal((M, K)).astype(np.float32), dev)
        z = dv(rng.standard_normal((M, N)).astype(np.float32), dev)
        if kind == "stream16":
            x, z = x.to(torch.bfloat16), z.to(torch.bfloat16)
            keep += [x, z]
        o0 = rng.standard_normal((K, N)).astype(np.float32)
        b0 = rng.standard_normal(N).astype(np.float32)

        def call(out, db, sc, kind=kind, M=M, K=K, N=N, acc=acc, x=x, z=z):
            if kind == "f32":
                return lib.dispu_linear_tn(1, M, K, N, p(x), K, 0, p(z), N, 0, p(out), N, 0, acc, p(db), p(sc), sc.numel(), st)
            if kind == "bf16":
                return lib.dispu_linear_tn_bf16(1, M, K, N, p(x), K, 0, p(z), N, 0, p(out), N, 0, acc, p(db), p(sc), sc.numel(), st)
            return lib.dispu_linear_tn_bf16_stream(M, K, N, p(x), K, p(z), N, 3 if kind == "stream16" else 0, p(out), N, acc, p(db), p(sc), sc.numel(), st)
        need = max({"f32": lib.dispu_linear_tn_scratch_floats(1, M, K, N), "bf16": lib.dispu_linear_tn_bf16_scratch_floats(1, M, K, N)}.get(
            kind, lib.dispu_linear_tn_bf16_stream_scratch_floats(M, K, N)), 4)
        res = []
        for deferred in (False, True):
            out, db = dv(o0, dev), (dv(b0, dev) if bias else None)
            sc = torch.empty(need, dtype=torch.float32, device=dev)
            keep.append(sc)
            if deferred:
                L.check(lib.dispu_tn_defer(C.c_void_p(C.addressof(descs) + i * C.sizeof(L.TnReduceDesc))), "defer")
            L.check(call(out, db, sc), kind)
            res.append((out, db))
        assert descs[i].splits > 1 and descs[i].K == K and descs[i].N == N, (kind, descs[i].splits)
        want.append(res[0])
        got.append(res[1])
    raw = C.string_at(C.addressof(descs), C.sizeof(descs))
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    for (o, b), (kind, M, K, N, acc, bias) in zip(got, jobs):          # nothing reduced yet: the destinations still hold their start values
        assert not torch.equal(o, want[jobs.index((kind, M, K, N, acc, bias))][0])
    L.check(lib.dispu_tn_reduce_grouped(len(jobs), C.c_void_p(C.addressof(descs)), C.c_void_p(table.data_ptr()), st), "grouped")
    torch.cuda.synchronize()
    for (wo, wb), (go, gb), job in zip(want, got, jobs):
        assert torch.equal(wo, go), job
        assert wb is None or torch.equal(wb, gb), job
    # one-shot: with nothing armed the product reduces itself; a batched product disarms the sink and reduces itself too
    d1 = (L.TnReduceDesc * 1)()
    x = dv(rng.standard_normal((2, 512, 64)).astype(np.float32), dev)
    z = dv(rng.standard_normal((2, 512, 32)).astype(np.float32), dev)
    o1, o2 = torch.zeros((2, 64, 32), device=dev), torch.zeros((2, 64, 32), device=dev)
    sc = torch.empty(max(lib.dispu_linear_tn_scratch_floats(2, 512, 64, 32), 4), dtype=torch.float32, device=dev)
    L.check(lib.dispu_linear_tn(2, 512, 64, 32, p(x), 64, 512 * 64, p(z), 32, 512 * 32, p(o1), 32, 64 * 32, 0, None, p(sc), sc.numel(), st), "tn")
    L.check(lib.dispu_tn_defer(C.c_void_p(C.addressof(d1))), "defer")
    L.check(lib.dispu_linear_tn(2, 512, 64, 32, p(x), 64, 512 * 64, p(z), 32, 512 * 32, p(o2), 32, 64 * 32, 0, None, p(sc), sc.numel(), st), "tn")
    torch.cuda.synchronize()
    assert d1[0].splits == 0 and torch.equal(o1, o2)
    close(N_(o1), np.einsum("bmk,bmn->bkn", N_(x).astype(np.float64), N_(z).astype(np.float64)), 1e-5, "batched TN")
    assert lib.dispu_tn_reduce_grouped(1, C.c_void_p(C.addressof(d1)), C.c_void_p(table.data_ptr()), st) != 0      # splits == 0: refused
    assert lib.dispu_tn_reduce_grouped(0, None, None, st) == 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_trainer_grouped_reductions_equal_single_ones(dev, dtype):
    """Trainer.group_reduce (the step's ~20 split reductions as one launch per stream) against every product reducing itself: the same
    gradients (float atomics elsewhere in the backward move the last bits: 1e-6 of the largest gradient), taped steps included."""
    from dispu_amd import synth
    from dispu_amd.params import init_params
    from dispu_amd.train import Trainer
    P = init_params(1234)
    x, gt = synth.patch_with_gt(8, 256, 1024, seed=11)
    x, gt = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev)
    radius = torch.ones(8, device=dev)
    g = {}
    for on in (False, True):
        tr = Trainer(params=P, device=dev, dtype=dtype)
        tr.group_reduce = on
        tr.zero_grad()
        tr.forward(x)
        tr.loss_backward(gt, radius)
        tr.backward()
        torch.cuda.synchronize()
        g[on] = tr.flat_g.clone()
        if on:
            assert len(tr._rg_dev) >= 1 and all(grp[1] == 0 for grp in tr._rg.values())
            for _ in range(3):
                terms = tr.train_step_taped(x, gt, radius)
            assert np.isfinite(float(terms["pu_loss"]))
    scale = float(g[False].abs().max())
    assert float((g[True] - g[False]).abs().max()) <= 5e-6 * scale


# ---------------------------------------------------------------------------------- round 6: local cell backward in one launch ----
@pytest.mark.parametrize("B,n", [(2, 256), (3, 64), (1, 1024)])
def test_ps_local_grad_equals_the_five_launch_path(dev, L, B, n):
    """dispu_ps_local_grad (csrc/ps_local_bwd.hip) against the launches it replaces, on the same inputs: gather_sub_relu -> dispu_linear
    (conv1) -> weight_net -> ps_point_matmul_grad_relu -> dispu_linear(transb) -> knn_invert + ps_conv0_gather_grad.  dz1 (recomputed conv1,
    its ReLU mask and the t-ascending contraction are the same arithmetic) bit-identical; dwv / dAneg re-associated sums and dG float atomics:
    1e-5 of the largest entry."""
    rng = np.random.default_rng(B * 1000 + n)
    k, c, t = 16, 128, 16
    xyz, idx = _cloud(rng, B, n, k)
    rows = B * n
    Gm = rng.standard_normal((rows, c)).astype(np.float32)
    Am = rng.standard_normal((rows, c)).astype(np.float32)
    W1 = (rng.standard_normal((c, c)) * 0.15).astype(np.float32)
    b1 = (rng.standard_normal(c) * 0.1).astype(np.float32)
    Ww, bw = rng.standard_normal((3, t)).astype(np.float32), rng.standard_normal(t).astype(np.float32)
    sc, sh = (1 + 0.1 * rng.standard_normal(t)).astype(np.float32), (0.1 * rng.standard_normal(t)).astype(np.float32)
    dF = rng.standard_normal((rows, c * t)).astype(np.float32)
    lib, st = L.lib(), L.stream_ptr(dev)
    di, dx = dv(idx, dev, torch.int32), dv(xyz.reshape(rows, 3), dev)
    tG, tA, tW1, tb1, tW1t = dv(Gm, dev), dv(Am, dev), dv(W1, dev), dv(b1, dev), dv(np.ascontiguousarray(W1.T), dev)
    tWw, tbw, tsc, tsh, tdF = dv(Ww, dev), dv(bw, dev), dv(sc, dev), dv(sh, dev), dv(dF, dev)
    E = lambda r, w: torch.empty((r, w), dtype=torch.float32, device=dev)
    # the five-launch path
    h0, h1, wv, dz1, dwv, dz0 = E(rows * k, c), E(rows * k, c), E(rows * k, t), E(rows * k, c), E(rows * k, t), E(rows * k, c)
    L.check(lib.dispu_ps_gather_sub_relu(rows, n, k, c, p(di), p(tG), c, p(tA), c, p(h0), c, st), "gather_sub_relu")
    L.check(lib.dispu_linear(1, rows * k, c, c, p(h0), c, 0, p(tW1), c, 0, 0, p(tb1), 1, p(h1), c, 0, None, 0, 0, None, 0, 0, st), "conv1")
    L.check(lib.dispu_ps_weight_net(rows, n, k, t, p(di), p(dx), p(tWw), p(tbw), p(tsc), p(tsh), p(wv), st), "weight_net")
    L.check(lib.dispu_ps_point_matmul_grad_relu(rows, k, c, t, p(h1), c, p(wv), p(tdF), c * t, p(dz1), c, p(dwv), st), "point_matmul_grad_relu")
    L.check(lib.dispu_linear(1, rows * k, c, c, p(dz1), c, 0, p(tW1), c, 0, 1, None, 0, p(dz0), c, 0, None, 0, 0, None, 0, 0, st), "conv1 dX")
    off = torch.empty((B, n + 1), dtype=torch.int32, device=dev)
    inv = torch.empty((B, n * k), dtype=torch.int32, device=dev)
    L.check(lib.dispu_knn_invert(B, n, k, p(di), p(off), p(inv), st), "knn_invert")
    dG, dA = E(rows, c), E(rows, c)
    L.check(lib.dispu_ps_conv0_gather_grad(rows, n, k, c, p(di), p(off), p(inv), p(dz0), c, p(tG), c, p(tA), c, p(dG), c, p(dA), c, st), "gather_grad")
    # one launch
    fz1, fwv, fG, fA = E(rows * k, c), E(rows * k, t), torch.zeros((rows, c), dtype=torch.float32, device=dev), E(rows, c)
    L.check(lib.dispu_ps_local_grad(rows, n, p(di), p(dx), p(tG), c, p(tA), p(tW1), p(tb1), p(tW1t), p(tWw), p(tbw), p(tsc), p(tsh), p(tdF),
                                    p(fz1), p(fwv), p(fG), p(fA), st), "ps_local_grad")
    torch.cuda.synchronize()
    _KEEP.extend([h0, h1, wv, dz1, dwv, dz0, off, inv, dG, dA, fz1, fwv, fG, fA])
    assert torch.equal(fz1, dz1), "dz1 differs from the unfused path"
    close(N_(fwv), N_(dwv).astype(np.float64), 1e-5, "dwv")
    close(N_(fG), N_(dG).astype(np.float64), 1e-5, "dG")
    close(N_(fA), N_(dA).astype(np.float64), 1e-5, "dAneg")
    assert lib.dispu_ps_local_grad(rows - 2, n, p(di), p(dx), p(tG), c, p(tA), p(tW1), p(tb1), p(tW1t), p(tWw), p(tbw), p(tsc), p(tsh), p(tdF),
                                   p(fz1), p(fwv), p(fG), p(fA), st) != 0           # whole 4-point groups only


def test_trainer_fused_local_backward_equals_the_unfused_one(dev):
    """Trainer.fused_local_bwd against the five-launch path inside the real step: the same gradients up to float atomics / re-associated
    sums (1e-5 of the largest gradient), and the side effects the unfused path has no more (h1, wv, the inverted graph) are not needed."""
    from dispu_amd import synth
    from dispu_amd.params import init_params
    from dispu_amd.train import Trainer
    P = init_params(1234)
    x, gt = synth.patch_with_gt(8, 256, 1024, seed=17)
    x, gt = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev)
    radius = torch.ones(8, device=dev)
    g = {}
    for on in (False, True):
        tr = Trainer(params=P, device=dev)
        tr.fused_local_bwd = on
        tr.zero_grad()
        tr.forward(x)
        tr.loss_backward(gt, radius)
        tr.backward()
        torch.cuda.synchronize()
        g[on] = tr.flat_g.clone()
        if on:
            for _ in range(2):
                terms = tr.train_step(x, gt, radius)
            assert np.isfinite(float(terms["pu_loss"]))
    scale = float(g[False].abs().max())
    assert float((g[True] - g[False]).abs().max()) <= 1e-5 * scale

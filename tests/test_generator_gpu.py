"""GPU parity of the generator forward (DisPU/generator.py:31-88 counterpart) against oracle/generator.py.
Everything that feeds an index decision (feature k-NN inside the dense blocks, xyz k-NN on `coarse`) is
bit-exact; the refinement branch (softmax / sigmoid / reassociated conv0) is checked at 1e-5."""
import numpy as np
import pytest
import torch

from oracle import generator as OG

pytestmark = pytest.mark.gpu


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def setup(dev):
    from dispu_amd import synth
    from dispu_amd.generator import Generator
    P = OG.init_params(seed=1234, bias_scale=0.05, bn_random=True)   # exercise biases and the BN fold
    gen = Generator(params=P, device=dev)
    gen.keep_intermediates = True                 # the fused fine head then also writes the aggregation output
    x = synth.patches(3, 256, seed=5)
    tap = {}
    coarse, fine = OG.generator_forward(P, x, tap)
    c, f = gen(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    return dict(gen=gen, P=P, x=x, tap=tap, coarse=coarse, fine=fine, c=N(c), f=N(f), ws=gen._ws[(3, 256)])


def test_feature_extraction_bit_exact(setup):
    ws, tap = setup["ws"], setup["tap"]
    assert np.array_equal(N(ws["kidx"])[:, 1:].reshape(3, 256, 16), tap["fe_idx4"])   # last dense block's neighbours
    assert np.array_equal(N(ws["feat"]).reshape(3, 256, 480), tap["feat480"])


def test_coarse_bit_exact(setup):
    assert np.array_equal(N(setup["ws"]["up128"]).reshape(3, 1024, 128), setup["tap"]["up128"])
    assert np.array_equal(setup["c"], setup["coarse"])


def test_pointshuffle_neighbours_index_exact(setup):
    assert np.array_equal(N(setup["ws"]["psidx"]).reshape(3, 1024, 16), setup["tap"]["ps_idx"])


def test_fine_within_tolerance(setup):
    ff = N(setup["ws"]["agg"]).reshape(3, 1024, 256)
    ref = setup["tap"]["fine_feat"]
    assert np.abs(ff - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(setup["f"] - setup["fine"]).max() <= 1e-5


def test_fused_local_cell_equals_unfused_chain(dev):
    """dispu_ps_local (one kernel) vs gather_sub_relu -> dispu_linear -> weight_net -> point_matmul: bit-identical."""
    from dispu_amd import synth
    from dispu_amd.generator import Generator
    P = OG.init_params(seed=7, bias_scale=0.05, bn_random=True)
    x = torch.from_numpy(synth.patches(2, 256, seed=11)).to(dev)
    outs = []
    for fused in (True, False):
        gen = Generator(params=P, device=dev)
        gen.fused_local = fused
        c, f = gen(x)
        outs.append((N(gen._ws[(2, 256)]["fp"]).copy(), N(c).copy(), N(f).copy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("clouds,n", [(1, 20), (3, 170), (2, 1024), (5, 256), (300, 24), (1200, 256)])
def test_ps_local_abi_ragged_groups(dev, clouds, n):
    """dispu_ps_local through the C ABI on point counts that are NOT a multiple of the kernel's 8-point groups (the last group is
    ragged: its missing points' stores are dropped by the buffer range check), on one group only, and on more groups than
    persistent workgroups (300 x 24 points = 900 groups over 256 workgroups), and on more than 2^18 points (1200 x 256: the
    entry point cuts the launch at a cloud boundary, the F' buffer resource has a 32-bit byte range) -- bit-identical to the unfused
    gather_sub_relu / dispu_linear / weight_net / point_matmul chain, and nothing written past the end of `out`."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(clouds * 1000 + n)
    npts = clouds * n
    xyz = torch.from_numpy(rng.random((npts, 3)).astype(np.float32)).to(dev)
    idx = torch.from_numpy(rng.integers(0, n, (npts, 16)).astype(np.int32)).to(dev)
    G = torch.from_numpy(rng.standard_normal((npts, 320)).astype(np.float32)).to(dev)
    A = torch.from_numpy(rng.standard_normal((npts, 128)).astype(np.float32)).to(dev)
    W1 = torch.from_numpy((rng.standard_normal((128, 128)) * 0.1).astype(np.float32)).to(dev)
    b1 = torch.from_numpy((rng.standard_normal(128) * 0.1).astype(np.float32)).to(dev)
    Ww = torch.from_numpy(rng.standard_normal((3, 16)).astype(np.float32)).to(dev)
    bw = torch.from_numpy(rng.standard_normal(16).astype(np.float32)).to(dev)
    sc = torch.from_numpy((1 + 0.1 * rng.standard_normal(16)).astype(np.float32)).to(dev)
    sh = torch.from_numpy((0.1 * rng.standard_normal(16)).astype(np.float32)).to(dev)
    st = _lib.stream_ptr(dev)
    P = lambda t, off=0: t.data_ptr() + 4 * off
    pad = 4096
    out = torch.full((npts * 2048 + pad,), -7.0, device=dev)                      # canary behind the result
    _lib.check(L.dispu_ps_local(npts, n, 16, 128, P(idx), P(xyz), P(G, 192), 320, P(A), P(W1), P(b1), P(Ww), P(bw), P(sc), P(sh),
                                P(out), st), "dispu_ps_local")
    x1 = torch.empty((npts * 16, 128), device=dev)
    x2 = torch.empty((npts * 16, 128), device=dev)
    wv = torch.empty((npts * 16, 16), device=dev)
    ref = torch.empty((npts, 2048), device=dev)
    _lib.check(L.dispu_ps_gather_sub_relu(npts, n, 16, 128, P(idx), P(G, 192), 320, P(A), 128, P(x1), 128, st), "gather_sub_relu")
    _lib.check(L.dispu_linear(1, npts * 16, 128, 128, P(x1), 128, 0, P(W1), 128, 0, 0, P(b1), 1, P(x2), 128, 0, None, 0, 0, None, 0, 0, st),
               "dispu_linear")
    _lib.check(L.dispu_ps_weight_net(npts, n, 16, 16, P(idx), P(xyz), P(Ww), P(bw), P(sc), P(sh), P(wv), st), "weight_net")
    _lib.check(L.dispu_ps_point_matmul(npts, 16, 128, 16, P(x2), 128, P(wv), P(ref), 2048, st), "point_matmul")
    got = out[:npts * 2048].view(npts, 2048)
    assert torch.equal(got, ref), int((got != ref).any(1).nonzero()[0])
    assert bool((out[npts * 2048:] == -7.0).all())


@pytest.mark.parametrize("b,m", [(2, 1024), (3, 160), (1, 4096)])
def test_fused_attention(dev, b, m):
    """dispu_attention (flash-style, logits on chip) vs a float64 softmax(QK^T/8)V and vs the 3-kernel path."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(m)
    q = rng.standard_normal((b, m, 64)).astype(np.float32)
    kv = rng.standard_normal((b, m, 128)).astype(np.float32)
    q[0, 5] *= 6.0                                              # a peaked row: exercises the running-max rescale
    tq, tkv = torch.from_numpy(q).to(dev), torch.from_numpy(kv).to(dev)
    out = torch.zeros((b, m, 64), device=dev)
    _lib.check(L.dispu_attention(b, m, m, 64, tq.data_ptr(), 64, tkv.data_ptr(), 128, tkv.data_ptr() + 256, 128, 0.125,
                                 out.data_ptr(), 64, _lib.stream_ptr(dev)), "dispu_attention")
    s = np.einsum("bqd,bkd->bqk", q.astype(np.float64), kv[..., :64].astype(np.float64)) / 8.0
    s -= s.max(-1, keepdims=True)
    p = np.exp(s)
    want = np.einsum("bqk,bkd->bqd", p / p.sum(-1, keepdims=True), kv[..., 64:].astype(np.float64))
    assert np.abs(N(out) - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


def test_fused_attention_path_equals_unfused_in_generator(dev):
    from dispu_amd import synth
    from dispu_amd.generator import Generator
    P = OG.init_params(seed=5, bias_scale=0.05)
    x = torch.from_numpy(synth.patches(2, 256, seed=13)).to(dev)
    res = []
    for fused in (True, False):
        gen = Generator(params=P, device=dev)
        gen.fused_attention = fused
        c, f = gen(x)
        res.append((N(gen._ws[(2, 256)]["nl"]).copy(), N(f).copy()))
    assert np.abs(res[0][0] - res[1][0]).max() <= 1e-5 * max(1.0, np.abs(res[1][0]).max())
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-6


def test_default_init_and_batch_independence(dev):
    """Xavier / zero-bias default init (the benchmark's weights); per-patch results do not depend on the batch."""
    from dispu_amd import synth
    from dispu_amd.generator import Generator
    P = OG.init_params()
    gen = Generator(params=P, device=dev)
    x = synth.patches(4, 256, seed=9)
    c4, f4 = gen(torch.from_numpy(x).to(dev))
    c4, f4 = N(c4).copy(), N(f4).copy()
    c1, f1 = gen(torch.from_numpy(x[2:3]).to(dev))
    assert np.array_equal(N(c1)[0], c4[2]) and np.array_equal(N(f1)[0], f4[2])
    co, fo = OG.generator_forward(P, x[2:3])
    assert np.array_equal(N(c1), co) and np.abs(N(f1) - fo).max() <= 1e-5


def test_linear_matches_chain_exactly(dev):
    """dispu_linear (fp32 MFMA) against the pinned fmaf chain for awkward shapes: K tail, N tail, M tail,
    strided operands, residuals, transposed B, batching."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for (M, K, Nn, act) in [(300, 134, 256, 1), (257, 482, 48, 0), (1024, 64, 64, 1), (70, 2048, 256, 1), (513, 120, 128, 0)]:
        x = rng.standard_normal((M, K + 3)).astype(np.float32)
        w = (rng.standard_normal((K, Nn)) * 0.1).astype(np.float32)
        b = rng.standard_normal(Nn).astype(np.float32)
        r1 = rng.standard_normal((M, Nn)).astype(np.float32)
        want = OG.linear(x[:, 1:K + 1], w, b, relu=bool(act)) + r1
        tx, tw, tb, tr = (torch.from_numpy(a).to(dev) for a in (x, w, b, r1))
        y = torch.zeros((M, Nn + 5), device=dev)
        _lib.check(L.dispu_linear(1, M, K, Nn, tx.data_ptr() + 4, K + 3, 0, tw.data_ptr(), Nn, 0, 0, tb.data_ptr(), act,
                                  y.data_ptr() + 8, Nn + 5, 0, tr.data_ptr(), Nn, 0, None, 0, 0,
                                  _lib.stream_ptr(dev)), "dispu_linear")
        got = N(y)
        assert np.array_equal(got[:, 2:Nn + 2], want), (M, K, Nn)
        assert (got[:, :2] == 0).all() and (got[:, Nn + 2:] == 0).all()
    q = rng.standard_normal((2, 96, 64)).astype(np.float32)
    kk = rng.standard_normal((2, 130, 64)).astype(np.float32)
    s = torch.empty((2, 96, 130), device=dev)
    tq, tk = torch.from_numpy(q).to(dev), torch.from_numpy(kk).to(dev)
    _lib.check(L.dispu_linear(2, 96, 64, 130, tq.data_ptr(), 64, 96 * 64, tk.data_ptr(), 64, 130 * 64, 1, None, 0,
                              s.data_ptr(), 130, 96 * 130, None, 0, 0, None, 0, 0, _lib.stream_ptr(dev)), "qk^t")
    assert np.array_equal(N(s), OG.matmul_nt(q, kk))


def test_linear_skinny_path_matches_chain_exactly(dev):
    """Small M x N (fewer than 256 tiles of 64 x 64), K <= 384, K % 4 == 0, aligned rows: dispu_linear runs the
    register-resident v_mfma_f32_16x16x4_f32 kernel (linear_skinny.hip) - same pinned ascending fmaf chain."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(7)
    for (M, K, Nn, act, ld, xo) in [(8192, 120, 48, 1, 480, 360), (8192, 240, 48, 1, 480, 240), (8192, 360, 48, 1, 480, 120),
                                    (1000, 124, 33, 0, 128, 4), (50, 16, 64, 1, 16, 0), (17, 4, 16, 0, 8, 4), (333, 384, 7, 1, 388, 4)]:
        x = rng.standard_normal((M, ld)).astype(np.float32)
        w = (rng.standard_normal((K, Nn)) * 0.1).astype(np.float32)
        b = rng.standard_normal(Nn).astype(np.float32)
        want = OG.linear(x[:, xo:xo + K], w, b, relu=bool(act))
        tx, tw, tb = (torch.from_numpy(a).to(dev) for a in (x, w, b))
        y = torch.zeros((M, Nn + 5), device=dev)
        _lib.check(L.dispu_linear(1, M, K, Nn, tx.data_ptr() + 4 * xo, ld, 0, tw.data_ptr(), Nn, 0, 0, tb.data_ptr(), act,
                                  y.data_ptr() + 8, Nn + 5, 0, None, 0, 0, None, 0, 0, _lib.stream_ptr(dev)), "dispu_linear")
        got = N(y)
        assert np.array_equal(got[:, 2:Nn + 2], want), (M, K, Nn)
        assert (got[:, :2] == 0).all() and (got[:, Nn + 2:] == 0).all()
        # no bias
        _lib.check(L.dispu_linear(1, M, K, Nn, tx.data_ptr() + 4 * xo, ld, 0, tw.data_ptr(), Nn, 0, 0, None, 0,
                                  y.data_ptr() + 8, Nn + 5, 0, None, 0, 0, None, 0, 0, _lib.stream_ptr(dev)), "dispu_linear")
        assert np.array_equal(N(y)[:, 2:Nn + 2], OG.linear(x[:, xo:xo + K], w, None, relu=False)), (M, K, Nn)


def test_linear_skinny_path_transb_and_residual(dev):
    """The skinny path also covers the training step's narrow shapes: outputs of <= 32 columns at any M, contractions of
    <= 32 (K % 4 == 0) with up to 128 columns, W given transposed ([N, K]) and a residual added after the activation."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(11)
    for (M, K, Nn, act, tb, res) in [(32768, 96, 24, 1, 0, 0), (32768, 24, 96, 0, 1, 1), (8192, 24, 72, 0, 1, 1), (4100, 16, 3, 0, 1, 1),
                                     (1000, 32, 100, 1, 1, 0), (999, 28, 128, 0, 0, 1), (5000, 120, 20, 1, 1, 1), (300, 384, 17, 0, 1, 0),
                                     (2048, 128, 134, 0, 1, 1), (700, 64, 262, 1, 0, 0), (1500, 256, 134, 0, 1, 0)]:   # 128 k + tail: split launches
        x = rng.standard_normal((M, K + 4)).astype(np.float32)
        w = (rng.standard_normal((K, Nn)) * 0.1).astype(np.float32)
        b = rng.standard_normal(Nn).astype(np.float32)
        r1 = rng.standard_normal((M, Nn)).astype(np.float32)
        want = OG.linear(x[:, 4:], w, b, relu=bool(act))
        if res:
            want = want + r1
        wt = np.ascontiguousarray(w.T) if tb else w
        tx, tw, tb_, tr = (torch.from_numpy(a).to(dev) for a in (x, wt, b, r1))
        y = torch.zeros((M, Nn + 3), device=dev)
        _lib.check(L.dispu_linear(1, M, K, Nn, tx.data_ptr() + 16, K + 4, 0, tw.data_ptr(), wt.shape[1], 0, tb, tb_.data_ptr(), act,
                                  y.data_ptr() + 4, Nn + 3, 0, tr.data_ptr() if res else None, Nn, 0, None, 0, 0,
                                  _lib.stream_ptr(dev)), "dispu_linear")
        got = N(y)
        assert np.array_equal(got[:, 1:Nn + 1], want), (M, K, Nn, tb, res)
        assert (got[:, :1] == 0).all() and (got[:, Nn + 1:] == 0).all()


# 2560 / 4096 points: more than 2 x 1024 waves of the persistent grid -> several point groups per wave (the weight
# fragment ring and the row prefetch wrap from one group into the next)
@pytest.mark.parametrize("C,npts", [(24, 512), (48, 512), (48, 777), (24, 2560), (48, 4096)])
def test_edge_dense_conv_mfma_equals_valu_and_oracle(dev, C, npts):
    """dense_conv on the matrix cores vs its VALU twin vs the oracle chain: all three bit-identical."""
    from dispu_amd import _lib
    from oracle import oracle as O
    L = _lib.lib()
    rng = np.random.default_rng(C + npts)
    n_cloud = npts if npts % 256 else 256
    nb = npts // n_cloud
    F = rng.standard_normal((nb, n_cloud, C)).astype(np.float32)
    _, idx2 = O.knn_point_2(17, F, F)
    idx = idx2[..., 1].astype(np.int32)                                     # [nb, n, 17]
    P = {k: (rng.standard_normal(s) * 0.2).astype(np.float32) for k, s in
         dict(W0=(2 * C, 24), b0=(24,), W1=(24 + C, 24), b1=(24,), W2=(48 + C, 24), b2=(24,)).items()}
    # oracle: the dense_conv restatement with these weights
    scope = "x"
    PP = {scope + "/l0/weights": P["W0"], scope + "/l0/biases": P["b0"], scope + "/l1/weights": P["W1"],
          scope + "/l1/biases": P["b1"], scope + "/l2/weights": P["W2"], scope + "/l2/biases": P["b2"]}
    want, widx = OG.dense_conv(PP, scope, F)
    assert np.array_equal(widx, idx[:, :, 1:])
    t = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}
    tF, tI = torch.from_numpy(F).to(dev), torch.from_numpy(idx).to(dev)
    outs = []
    for fn in (L.dispu_edge_dense_conv, L.dispu_edge_dense_conv_valu):
        y = torch.zeros((npts, 72 + C + 3), device=dev)
        _lib.check(fn(npts, n_cloud, C, tF.data_ptr(), C, tI.data_ptr(), 17, 1, t["W0"].data_ptr(), t["b0"].data_ptr(),
                      t["W1"].data_ptr(), t["b1"].data_ptr(), t["W2"].data_ptr(), t["b2"].data_ptr(), y.data_ptr(), 72 + C + 3,
                      _lib.stream_ptr(dev)), "edge")
        outs.append(N(y))
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0][:, :72 + C].reshape(nb, n_cloud, 72 + C), want)
    assert (outs[0][:, 72 + C:] == 0).all()


@pytest.mark.parametrize("C,nb,n_cloud,kind", [(24, 32, 256, "normal"), (48, 32, 256, "normal"), (48, 1, 256, "normal"), (48, 5, 256, "normal"),
                                                (48, 8, 256, "normal"), (24, 3, 128, "normal"), (48, 70, 256, "normal"), (48, 2, 64, "normal"),
                                                (48, 3, 34, "normal"), (48, 2, 256, "dups"), (24, 2, 256, "grid"), (48, 2, 250, "normal")])
def test_stem_block_equals_search_then_edge_conv(dev, C, nb, n_cloud, kind):
    """dispu_stem_block (neighbour search + edge features + dense_conv in one launch) vs dispu_knn_feat_strided followed by
    dispu_edge_dense_conv: neighbour table and output bit-identical; vs the oracle on the small cases.  'dups': clouds made of 8
    distinct rows (every distance a 32-way tie: the sort-everything path), 'grid': features on a coarse lattice (many exact ties)."""
    from dispu_amd import _lib
    from oracle import oracle as O
    L = _lib.lib()
    rng = np.random.default_rng(C + nb + n_cloud)
    F = rng.standard_normal((nb, n_cloud, C)).astype(np.float32)
    if kind == "dups":
        F = F[:, rng.integers(0, 8, n_cloud)]
    elif kind == "grid":
        F = np.round(F * 2) / 2
    npts = nb * n_cloud
    ld = C + 8                                              # strided input rows
    Fp = np.zeros((npts, ld), np.float32)
    Fp[:, :C] = F.reshape(npts, C)
    P = {k: (rng.standard_normal(s) * 0.2).astype(np.float32) for k, s in
         dict(W0=(2 * C, 24), b0=(24,), W1=(24 + C, 24), b1=(24,), W2=(48 + C, 24), b2=(24,)).items()}
    t = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}
    tF = torch.from_numpy(Fp).to(dev)
    st = _lib.stream_ptr(dev)
    w = [t[k].data_ptr() for k in ("W0", "b0", "W1", "b1", "W2", "b2")]
    idx_a = torch.full((npts, 17), -1, dtype=torch.int32, device=dev)
    y_a = torch.zeros((npts, 72 + C + 3), device=dev)
    _lib.check(L.dispu_knn_feat_strided(nb, n_cloud, n_cloud, C, 17, tF.data_ptr(), ld, tF.data_ptr(), ld, None, idx_a.data_ptr(), st), "knn")
    _lib.check(L.dispu_edge_dense_conv(npts, n_cloud, C, tF.data_ptr(), ld, idx_a.data_ptr(), 17, 1, *w, y_a.data_ptr(), 72 + C + 3, st), "edge")
    for with_idx in (True, False):
        idx_b = torch.full((npts, 17), -1, dtype=torch.int32, device=dev)
        y_b = torch.zeros((npts, 72 + C + 3), device=dev)
        # with_idx also exercises the bottleneck-conv epilogue: y rows continue with k_old older columns
        k_old = 24 * (1 + (nb % 3)) if with_idx else 0
        ldy = 72 + C + k_old + (4 if with_idx else 3)
        y_b = torch.zeros((npts, ldy), device=dev)
        y_b[:, 72 + C:] = torch.from_numpy(rng.standard_normal((npts, ldy - 72 - C)).astype(np.float32)).to(dev)
        tail = N(y_b)[:, 72 + C:].copy()
        Wp = torch.from_numpy((rng.standard_normal((72 + C + k_old, 48)) * 0.1).astype(np.float32)).to(dev)
        bp = torch.from_numpy((rng.standard_normal((48,)) * 0.1).astype(np.float32)).to(dev)
        pr = torch.full((npts, 50), -7.0, device=dev)
        _lib.check(L.dispu_stem_block(npts, n_cloud, C, tF.data_ptr(), ld, 17, 1, *w, y_b.data_ptr(), ldy,
                                      idx_b.data_ptr() if with_idx else None, Wp.data_ptr() if with_idx else None,
                                      bp.data_ptr() if with_idx else None, k_old, pr.data_ptr() if with_idx else None, 50, None, None, None, None, 0,
                                      st), "stem")
        assert np.array_equal(N(y_a)[:, :72 + C], N(y_b)[:, :72 + C])
        assert np.array_equal(N(y_b)[:, 72 + C:], tail)
        if with_idx:
            want = torch.zeros((npts, 48), device=dev)
            _lib.check(L.dispu_linear(1, npts, 72 + C + k_old, 48, y_b.data_ptr(), ldy, 0, Wp.data_ptr(), 48, 0, 0, bp.data_ptr(), 1,
                                      want.data_ptr(), 48, 0, None, 0, 0, None, 0, 0, st), "linear")
            assert np.array_equal(N(pr)[:, :48], N(want))
            assert (N(pr)[:, 48:] == -7.0).all()
        else:
            assert (N(pr) == -7.0).all()
        if with_idx:
            assert np.array_equal(N(idx_a), N(idx_b))
        else:
            assert (N(idx_b) == -1).all()
    if npts <= 2048:
        _, idx2 = O.knn_point_2(17, F, F)
        assert np.array_equal(idx2[..., 1].reshape(npts, 17), N(idx_a))


@pytest.mark.parametrize("nb,n_cloud", [(32, 256), (3, 128), (5, 250)])
def test_stem_block_with_layer0_from_coordinates(dev, nb, n_cloud):
    """First dense block fed with coordinates (layer0 evaluated while the cloud is staged) vs dispu_linear_small_k followed by the
    block on its output: layer0 rows, block output, neighbour table and the next block's bottleneck conv bit-identical."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(nb + n_cloud)
    npts = nb * n_cloud
    xyz = torch.from_numpy(rng.uniform(-1, 1, (npts, 3)).astype(np.float32)).to(dev)
    Wl = torch.from_numpy((rng.standard_normal((3, 24)) * 0.5).astype(np.float32)).to(dev)
    bl = torch.from_numpy((rng.standard_normal((24,)) * 0.1).astype(np.float32)).to(dev)
    P = {k: (rng.standard_normal(s) * 0.2).astype(np.float32) for k, s in
         dict(W0=(48, 24), b0=(24,), W1=(48, 24), b1=(24,), W2=(72, 24), b2=(24,)).items()}
    t = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}
    w = [t[k].data_ptr() for k in ("W0", "b0", "W1", "b1", "W2", "b2")]
    Wp = torch.from_numpy((rng.standard_normal((120, 48)) * 0.1).astype(np.float32)).to(dev)
    bp = torch.from_numpy((rng.standard_normal((48,)) * 0.1).astype(np.float32)).to(dev)
    st = _lib.stream_ptr(dev)
    outs = []
    for fused in (False, True):
        feat = torch.zeros((npts, 480), device=dev)                # the generator's layout: layer0 at 456:480, the block at 360:456
        idx = torch.full((npts, 17), -1, dtype=torch.int32, device=dev)
        pr = torch.zeros((npts, 48), device=dev)
        off = lambda c: feat.data_ptr() + 4 * c
        if not fused:
            _lib.check(L.dispu_linear_small_k(npts, 3, 24, xyz.data_ptr(), 3, Wl.data_ptr(), bl.data_ptr(), 0, off(456), 480, st), "layer0")
        _lib.check(L.dispu_stem_block(npts, n_cloud, 24, None if fused else off(456), 480, 17, 1, *w, off(360), 480, idx.data_ptr(), Wp.data_ptr(),
                                      bp.data_ptr(), 24, pr.data_ptr(), 48, xyz.data_ptr() if fused else None, Wl.data_ptr() if fused else None,
                                      bl.data_ptr() if fused else None, off(456) if fused else None, 480, st), "stem")
        outs.append((N(feat).copy(), N(idx).copy(), N(pr).copy()))
    for a_, b_ in zip(*outs):
        assert np.array_equal(a_, b_)
    assert np.abs(outs[0][0][:, 456:]).max() > 0 and np.abs(outs[0][2]).max() > 0


def test_stem_block_refuses_what_it_does_not_cover(dev):
    from dispu_amd import _lib
    L = _lib.lib()
    x = torch.zeros((1024, 48), device=dev)
    w = torch.zeros((4096,), device=dev)
    y = torch.zeros((1024, 144), device=dev)
    st = _lib.stream_ptr(dev)
    a = lambda npts, n, C, ksel, ioff, k_old=0, wp=None: L.dispu_stem_block(npts, n, C, x.data_ptr(), 48, ksel, ioff, *([w.data_ptr()] * 6), y.data_ptr(),
                                                                            120 + 24, None, wp, wp, k_old, wp, 48, None, None, None, None, 0, st)
    assert a(1024, 512, 48, 17, 1) != 0            # clouds of more than 256 points
    assert a(1024, 256, 32, 17, 1) != 0            # C outside {24, 48}
    assert a(1000, 256, 48, 17, 1) != 0            # ragged clouds
    assert a(1024, 256, 48, 18, 1) != 0            # ksel != ioff + 16
    assert a(1024, 16, 48, 17, 1) != 0             # fewer points than neighbours
    assert a(1024, 256, 48, 17, 1, 20, w.data_ptr()) != 0   # older columns not a multiple of 24
    assert a(1024, 256, 48, 17, 1) == 0


def test_fused_head_chains_equal_separate_launches(dev):
    """dispu_mlp_chain (one launch per head, activations in LDS) vs the dispu_linear / dispu_linear_small_n launches:
    bit-identical coarse, fine, up128 and aggregation output."""
    from dispu_amd import synth
    from dispu_amd.generator import Generator
    P = OG.init_params(seed=3, bias_scale=0.05, bn_random=True)
    x = torch.from_numpy(synth.patches(2, 256, seed=13)).to(dev)
    outs = []
    for fused in (True, False):
        gen = Generator(params=P, device=dev)
        gen.fused_heads = fused
        gen.keep_intermediates = True
        c, f = gen(x)
        ws = gen._ws[(2, 256)]
        outs.append([N(c).copy(), N(f).copy(), N(ws["up128"]).copy(), N(ws["agg"]).copy()])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("B", [1, 5])
def test_odd_batch_sizes_match_oracle(dev, B):
    """BASELINE configs[0] (a single 256-point patch) and a batch that is not a multiple of anything: persistent /
    per-cloud kernels must cope with fewer point groups than CUs and with ragged grids."""
    from dispu_amd import synth
    from dispu_amd.generator import Generator
    P = OG.init_params(seed=11, bias_scale=0.05, bn_random=True)
    x = synth.patches(B, 256, seed=40 + B)
    coarse, fine = OG.generator_forward(P, x)
    c, f = Generator(params=P, device=dev)(torch.from_numpy(x).to(dev))
    assert np.array_equal(N(c), coarse)
    assert np.abs(N(f) - fine).max() <= 1e-5


@pytest.mark.parametrize("b,m", [(2, 1024), (3, 160), (1, 4096)])
def test_fused_attention_project(dev, b, m):
    """dispu_attention_project = relu(softmax(QK^T/8) V W + bias) (PointNonLocalCell incl. conv_back_project, ops.py:326-343)
    vs float64, and vs dispu_attention followed by dispu_linear (the unfused pair)."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(m + 1)
    q = rng.standard_normal((b, m, 64)).astype(np.float32)
    kv = rng.standard_normal((b, m, 128)).astype(np.float32)
    w = (rng.standard_normal((64, 256)) * 0.2).astype(np.float32)
    bias = (rng.standard_normal(256) * 0.1).astype(np.float32)
    q[0, 7] *= 5.0
    tq, tkv, tw, tb = (torch.from_numpy(a).to(dev) for a in (q, kv, w, bias))
    y = torch.zeros((b, m, 256), device=dev)
    st = _lib.stream_ptr(dev)
    _lib.check(L.dispu_attention_project(b, m, m, 64, tq.data_ptr(), 64, tkv.data_ptr(), 128, tkv.data_ptr() + 256, 128, 0.125,
                                         tw.data_ptr(), tb.data_ptr(), 256, y.data_ptr(), 256, st), "dispu_attention_project")
    s = np.einsum("bqd,bkd->bqk", q.astype(np.float64), kv[..., :64].astype(np.float64)) / 8.0
    s -= s.max(-1, keepdims=True)
    p = np.exp(s)
    att = np.einsum("bqk,bkd->bqd", p / p.sum(-1, keepdims=True), kv[..., 64:].astype(np.float64))
    want = np.maximum(att @ w.astype(np.float64) + bias, 0.0)
    assert np.abs(N(y) - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    o = torch.zeros((b, m, 64), device=dev)
    _lib.check(L.dispu_attention(b, m, m, 64, tq.data_ptr(), 64, tkv.data_ptr(), 128, tkv.data_ptr() + 256, 128, 0.125,
                                 o.data_ptr(), 64, st), "dispu_attention")
    y2 = torch.zeros((b * m, 256), device=dev)
    _lib.check(L.dispu_linear(1, b * m, 64, 256, o.data_ptr(), 64, 0, tw.data_ptr(), 256, 0, 0, tb.data_ptr(), 1, y2.data_ptr(), 256, 0,
                              None, 0, 0, None, 0, 0, st), "dispu_linear")
    assert np.abs(N(y).reshape(-1, 256) - N(y2)).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_projection_epilogue_equals_separate_gemm_in_generator(dev):
    from dispu_amd import synth
    from dispu_amd.generator import Generator
    P = OG.init_params(seed=5, bias_scale=0.05)
    x = torch.from_numpy(synth.patches(2, 256, seed=13)).to(dev)
    res = []
    for fused in (True, False):
        gen = Generator(params=P, device=dev)
        gen.fused_project = fused
        c, f = gen(x)
        res.append((N(gen._ws[(2, 256)]["nl"]).copy(), N(f).copy()))
    assert np.abs(res[0][0] - res[1][0]).max() <= 1e-5 * max(1.0, np.abs(res[1][0]).max())
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-6

"""GPU parity of the compositions: PointNet++ SA / MSG / FP modules, EdgeConv + kNN graph,
Chamfer / Hausdorff / EMD / repulsion losses -- dis-pu_amd/{pointnet_util,gcn_lib,loss_utils}.py vs oracle/modules.py.
Index outputs are exact; features go through fp32 GEMM chains that are bit-equal to the oracle's fmaf chains, the
pooling / BN / loss reductions are compared at 1e-5."""
import numpy as np
import pytest
import torch

from oracle import modules as OM

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def make_params(rng, spec, bn):
    """spec: list of (scope, cin, cout)."""
    P = {}
    for scope, cin, cout in spec:
        P[scope + "/weights"] = (rng.standard_normal((cin, cout)) * (1.0 / np.sqrt(cin))).astype(np.float32)
        P[scope + "/biases"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        if bn:
            P[scope + "/bn/gamma"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
            P[scope + "/bn/beta"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
            P[scope + "/bn/moving_mean"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
            P[scope + "/bn/moving_variance"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    return P


def cloud(rng, b, n):
    return rng.random((b, n, 3)).astype(np.float32)


@pytest.mark.parametrize("pooling", ["max", "avg", "min", "weighted_avg", "max_and_avg"])
@pytest.mark.parametrize("knn", [False, True])
def test_pointnet_sa_module(dev, pooling, knn):
    from dispu_amd import pointnet_util as PU
    rng = np.random.default_rng(1)
    xyz, pts = cloud(rng, 2, 512), rng.standard_normal((2, 512, 13)).astype(np.float32)
    c_pool = 64 if pooling == "max_and_avg" else 32
    P = make_params(rng, [("sa/conv0", 16, 24), ("sa/conv1", 24, 32), ("sa/conv_post_0", c_pool, 40)], bn=True)
    want_xyz, want_pts, want_idx = OM.pointnet_sa_module(P, "sa", xyz, pts, 96, 0.2, 16, [24, 32], [40], False, bn=True,
                                                         pooling=pooling, knn=knn)
    got_xyz, got_pts, got_idx = PU.pointnet_sa_module(T(xyz, dev), T(pts, dev), 96, 0.2, 16, [24, 32], [40], False, False,
                                                      None, "sa", bn=True, pooling=pooling, knn=knn, params=P)
    assert np.array_equal(N(got_xyz), want_xyz) and np.array_equal(N(got_idx), want_idx)
    assert np.allclose(N(got_pts), want_pts, rtol=1e-5, atol=1e-5)


def test_pointnet_sa_group_all_and_msg(dev):
    from dispu_amd import pointnet_util as PU
    rng = np.random.default_rng(2)
    xyz, pts = cloud(rng, 2, 256), rng.standard_normal((2, 256, 8)).astype(np.float32)
    P = make_params(rng, [("ga/conv0", 11, 32), ("ga/conv1", 32, 64)], bn=False)
    wx, wp, wi = OM.pointnet_sa_module(P, "ga", xyz, pts, None, None, None, [32, 64], None, True, bn=False)
    gx, gp, gi = PU.pointnet_sa_module(T(xyz, dev), T(pts, dev), None, None, None, [32, 64], None, True, False, None, "ga",
                                       bn=False, params=P)
    assert N(gp).shape == (2, 1, 64) and np.array_equal(N(gi), wi) and np.array_equal(N(gx), wx)
    assert np.array_equal(N(gp), wp)                       # GEMM chain + max pooling: bit-exact
    Pm = make_params(rng, [("msg/conv0_0", 11, 16), ("msg/conv0_1", 16, 32), ("msg/conv1_0", 11, 24)], bn=True)
    wx, wp = OM.pointnet_sa_module_msg(Pm, "msg", xyz, pts, 64, [0.1, 0.3], [8, 24], [[16, 32], [24]])
    gx, gp = PU.pointnet_sa_module_msg(T(xyz, dev), T(pts, dev), 64, [0.1, 0.3], [8, 24], [[16, 32], [24]], params=Pm)
    assert np.array_equal(N(gx), wx) and N(gp).shape == (2, 64, 56) and np.allclose(N(gp), wp, rtol=1e-5, atol=1e-5)


def test_pointnet_fp_module(dev):
    from dispu_amd import pointnet_util as PU
    rng = np.random.default_rng(3)
    xyz1, xyz2 = cloud(rng, 2, 384), cloud(rng, 2, 128)
    p1, p2 = rng.standard_normal((2, 384, 10)).astype(np.float32), rng.standard_normal((2, 128, 20)).astype(np.float32)
    P = make_params(rng, [("fp/conv_0", 30, 48), ("fp/conv_1", 48, 32)], bn=True)
    want = OM.pointnet_fp_module(P, "fp", xyz1, xyz2, p1, p2, [48, 32], bn=True)
    got = PU.pointnet_fp_module(T(xyz1, dev), T(xyz2, dev), T(p1, dev), T(p2, dev), [48, 32], False, None, "fp", bn=True, params=P)
    assert np.allclose(N(got), want, rtol=1e-5, atol=1e-5)
    want = OM.pointnet_fp_module(make_params(rng, [("f2/conv_0", 20, 16)], False) if False else P, "fp", xyz1, xyz2, p1, p2, [48, 32])
    assert want.shape == (2, 384, 32)


def test_edge_conv_and_knn_graph(dev):
    from dispu_amd import gcn_lib as GL
    rng = np.random.default_rng(4)
    f = rng.standard_normal((3, 200, 20)).astype(np.float32)
    P = make_params(rng, [("gcn/ec", 40, 64)], bn=False)
    idx = OM.knn_graph(f, 16)
    got_idx = GL.knn_graph(T(f, dev), 16)
    assert np.array_equal(N(got_idx), idx) and np.array_equal(idx[:, :, 0], np.broadcast_to(np.arange(200), (3, 200)))
    want = OM.edge_conv_layer(P, "gcn/ec", f, idx)
    got = GL.edge_conv_layer(T(f, dev).unsqueeze(2), got_idx, 16, 64, scope="gcn/ec", params=P)
    assert N(got).shape == (3, 200, 1, 64) and np.array_equal(N(got), want)


def test_mrgcn_graphsage_gin_layers(dev):
    """The other three DeepGCN vertex layers the reference defines (gcn_lib/tf_vertex.py:20-79, 103-180, 182-251)."""
    from dispu_amd import gcn_lib as GL
    rng = np.random.default_rng(14)
    f = rng.standard_normal((2, 150, 12)).astype(np.float32)
    P = make_params(rng, [("g/mr", 24, 32), ("g/sage_aggr", 12, 12), ("g/sage", 24, 40), ("g/gin", 12, 20)], bn=False)
    P["g/gin_epsilon"] = np.array([0.25], np.float32)
    idx = OM.knn_graph(f, 9)
    tidx, tf_ = T(idx.astype(np.int32), dev), T(f, dev).unsqueeze(2)
    got = GL.max_relat_conv_layer(tf_, tidx, 9, 32, scope="g/mr", params=P)
    assert N(got).shape == (2, 150, 1, 32) and np.array_equal(N(got), OM.max_relat_conv_layer(P, "g/mr", f, idx))
    got = GL.graphsage_conv_layer(tf_, tidx, 9, 40, normalize=False, scope="g/sage", params=P)
    assert np.array_equal(N(got), OM.graphsage_conv_layer(P, "g/sage", f, idx, normalize=False))
    got = GL.graphsage_conv_layer(tf_, tidx, 9, 40, normalize=True, scope="g/sage", params=P)
    want = OM.graphsage_conv_layer(P, "g/sage", f, idx, normalize=True)
    assert np.allclose(N(got), want, rtol=1e-6, atol=1e-7)
    nrm = np.sqrt((N(got).astype(np.float64) ** 2).sum(-1))
    assert np.all((np.abs(nrm - 1) < 1e-5) | (nrm == 0))
    got = GL.gin_conv_layer(tf_, tidx, 9, 20, scope="g/gin", params=P)
    assert N(got).shape == (2, 150, 1, 20) and np.array_equal(N(got), OM.gin_conv_layer(P, "g/gin", f, idx))


def test_losses(dev):
    from dispu_amd import loss_utils as LU
    from dispu_amd import synth
    pred, gt = synth.patch_with_gt(3, 512, 512, seed=5)
    tp, tg = T(pred, dev).requires_grad_(True), T(gt, dev)
    cd = LU.chamfer(tp, tg, radius=1.0)
    assert abs(float(cd) - OM.chamfer(pred, gt)) <= 1e-5 * max(1.0, abs(OM.chamfer(pred, gt)))
    assert abs(float(LU.hausdorff_loss(tp, tg)) - OM.hausdorff_loss(pred, gt)) <= 1e-6
    emd = LU.earth_mover(tp, tg)
    assert abs(float(emd) - OM.earth_mover(pred, gt)) <= 1e-5 * OM.earth_mover(pred, gt)
    rep = LU.get_repulsion_loss(tp)
    assert abs(float(rep) - OM.get_repulsion_loss(pred)) <= 1e-6
    # gradients: chamfer through nn_distance_grad == finite differences of the oracle loss
    cd.backward()
    g = N(tp.grad)
    eps = 1e-3
    for (b, i, c) in [(0, 3, 0), (1, 100, 2), (2, 511, 1)]:
        e = np.zeros_like(pred); e[b, i, c] = eps
        num = (OM.chamfer(pred + e, gt) - OM.chamfer(pred - e, gt)) / (2 * eps)
        assert abs(num - g[b, i, c]) < 2e-4
    tp.grad = None
    emd.backward()
    assert np.isfinite(N(tp.grad)).all() and np.abs(N(tp.grad)).max() > 0


def test_hausdorff_loss_gradient(dev):
    """hausdorff_loss (Common/loss_utils.py:67-84) is logged, not trained on, in the reference; its gradient here flows through
    nn_distance's registered gradient and the two max reductions.  Checked against finite differences of the oracle loss at the
    coordinates that carry it (the points realising the maxima) and zero elsewhere."""
    from dispu_amd import loss_utils as LU
    from dispu_amd import synth
    pred, gt = synth.patch_with_gt(2, 300, 280, seed=9)
    tp, tg = T(pred, dev).requires_grad_(True), T(gt, dev)
    hd = LU.hausdorff_loss(tp, tg)
    hd.backward()
    g = N(tp.grad)
    assert abs(float(hd) - OM.hausdorff_loss(pred, gt)) <= 1e-6
    nz = np.argwhere(np.abs(g).sum(-1) > 0)
    assert 1 <= len(nz) <= 4                                # at most the two arg-max points of the winning cloud (+ their matches)
    eps = 1e-3
    for (b, i) in nz:
        for c in range(3):
            e = np.zeros_like(pred); e[b, i, c] = eps
            num = (OM.hausdorff_loss(pred + e, gt) - OM.hausdorff_loss(pred - e, gt)) / (2 * eps)
            assert abs(num - g[b, i, c]) < 5e-3 * max(1.0, abs(num))


# ---- round 4: the set-abstraction hot loop as one kernel (csrc/sa_fused.hip) and Common/ops.py:505-550 at its own shapes ----------
@pytest.mark.parametrize("c,mlp,ns,bn,knn", [(0, [32, 32, 64], 64, True, False), (64, [64, 64, 128], 64, True, False),
                                             (128, [128, 128, 256], 64, True, False), (13, [33, 17, 40], 64, False, False),
                                             (5, [24, 48], 32, True, True), (0, [16], 32, False, False), (260, [64, 64], 64, True, False)])
def test_sa_fused_equals_unfused_chain(dev, monkeypatch, c, mlp, ns, bn, knn):
    """pointnet_sa_module through dispu_sa_fused (group -> centre -> MLP -> max in one launch, nothing of size [b, m, ns, C] in HBM)
    == the composition group_point / group_center / linear_bn x nl / pool_nsample BIT FOR BIT, and the oracle to 1e-5; odd input and
    hidden widths (67, 131, 33, 17), 32 and 64 samples, with / without BatchNorm, ball query and k-NN grouping."""
    from dispu_amd import pointnet_util as PU
    rng = np.random.default_rng(c * 7 + ns)
    b, n, m = 3, 512, 96
    xyz = cloud(rng, b, n)
    pts = rng.standard_normal((b, n, c)).astype(np.float32) if c else None
    spec, cin = [], 3 + c
    for i, co in enumerate(mlp):
        spec.append(("sa/conv%d" % i, cin, co))
        cin = co
    P = make_params(rng, spec, bn=bn)
    tp = T(pts, dev) if c else None
    calls = []
    real = PU._sa_fused
    monkeypatch.setattr(PU, "_sa_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    fx, fp, fi = PU.pointnet_sa_module(T(xyz, dev), tp, m, 0.25, ns, mlp, None, False, False, None, "sa", bn=bn, knn=knn, params=P)
    assert calls, "the fused kernel was not used"
    monkeypatch.setattr("dispu_amd.pointnet_util.FUSED_SA", False)
    ux, up, ui = PU.pointnet_sa_module(T(xyz, dev), tp, m, 0.25, ns, mlp, None, False, False, None, "sa", bn=bn, knn=knn, params=P)
    assert len(calls) == 1
    assert np.array_equal(N(fi), N(ui)) and np.array_equal(N(fx), N(ux))
    assert N(fp).shape == (b, m, mlp[-1]) and np.array_equal(N(fp), N(up)), "fused != unfused chain (bit-exact expected)"
    wx, wp, wi = OM.pointnet_sa_module(P, "sa", xyz, pts, m, 0.25, ns, mlp, None, False, bn=bn, knn=knn)
    assert np.array_equal(N(fi), wi) and np.allclose(N(fp), wp, rtol=1e-5, atol=1e-5)


def test_hierachy_feature_extractor_at_reference_shapes(dev, monkeypatch):
    """Common/ops.py:505-550 assembled from the modules at the shapes the reference names: B = 4 clouds of 1024 points, levels of
    1024 / 384 / 128 centres with 64 samples and radii .1 / .2 / .4, MLPs up to 512 wide, group-all, four FP levels -- against
    oracle/modules.py.  The sampled levels run fused; the unfused composition gives the same bits."""
    from dispu_amd import ops as DO
    rng = np.random.default_rng(11)
    P = make_params(rng, DO.hierachy_feature_extractor_variables(), bn=True)
    x = cloud(rng, 4, 1024)
    want = OM.hierachy_feature_extractor(P, x)
    got = DO.hierachy_feature_extractor(T(x, dev), False, params=P)
    assert N(got).shape == (4, 1024, 128)
    scale = np.abs(want).max()
    assert np.abs(N(got) - want).max() <= 1e-5 * max(scale, 1.0), np.abs(N(got) - want).max()
    monkeypatch.setattr("dispu_amd.pointnet_util.FUSED_SA", False)
    unf = DO.hierachy_feature_extractor(T(x, dev), False, params=P)
    assert np.array_equal(N(unf), N(got))
    # the variable inventory is the checkpoint's: 21 conv layers with their BatchNorm quartets
    assert len(DO.hierachy_feature_extractor_variables()) == 21 and sum(k.endswith("/weights") for k in P) == 21


def test_sa_fused_refuses_bad_arguments(dev):
    from dispu_amd import _lib
    z = torch.zeros(64, device=dev)
    bad = _lib.lib().dispu_sa_fused(1, 8, 1, 48, 0, _lib.ptr(z), _lib.ptr(z), None, _lib.ptr(z), 1, None, None, None, None, None, _lib.ptr(z),
                                    _lib.stream_ptr(dev))
    assert bad != 0


@pytest.mark.parametrize("b,n,c,k,co,bn,act", [(3, 200, 20, 16, 64, False, "relu"), (2, 1024, 64, 16, 128, True, "relu"), (5, 33, 7, 16, 33, True, None),
                                               (2, 256, 128, 32, 256, False, "relu"), (1, 70, 24, 64, 48, True, "relu"), (4, 1024, 256, 16, 128, False, "relu")])
def test_edge_conv_fused_equals_unfused(dev, monkeypatch, b, n, c, k, co, bn, act):
    """EdgeConv (gcn_lib/tf_vertex.py:81-101: get_edge_feature -> conv2d -> max over k) as ONE launch (dispu_edge_conv_fused) == the
    three-launch composition bit for bit, and == oracle/modules.py; k = 16 / 32 / 64, odd widths, partial last workgroup (n = 33, 70),
    with / without BatchNorm and activation, c up to 256 (2c = 512-wide rows in LDS)."""
    from dispu_amd import gcn_lib as GL
    rng = np.random.default_rng(b * 100 + c)
    f = rng.standard_normal((b, n, c)).astype(np.float32)
    P = make_params(rng, [("ec", 2 * c, co)], bn=bn)
    if n >= k:
        idx = np.stack([np.stack([rng.permutation(n)[:k] for _ in range(n)]) for _ in range(b)]).astype(np.int32)
    else:
        idx = rng.integers(0, n, (b, n, k)).astype(np.int32)
    tf_, ti = T(f, dev).unsqueeze(2), T(idx, dev)
    got = GL.edge_conv_layer(tf_, ti, k, co, scope="ec", params=P, bn=bn, activation_fn=act)
    monkeypatch.setattr("dispu_amd.gcn_lib.FUSED_EDGE_CONV", False)
    unf = GL.edge_conv_layer(tf_, ti, k, co, scope="ec", params=P, bn=bn, activation_fn=act)
    assert N(got).shape == (b, n, 1, co) and np.array_equal(N(got), N(unf)), "fused EdgeConv != unfused composition"
    want = OM.edge_conv_layer(P, "ec", f, idx, bn=bn, relu=act == "relu")
    assert np.allclose(N(got), want, rtol=1e-5, atol=1e-5)

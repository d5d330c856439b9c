"""GPU parity tests (pytest -m gpu): every HIP op, called through the reference-named Python shims ->
ctypes -> the C ABI of libdispu_hip.so, against (a) the golden vectors produced by the reference's own
CPU functions, (b) the CPU oracle on seeded inputs.  Indices / gathers / plain-arithmetic distances are
bit-exact; approxmatch & reductions use the stated tolerances."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

PLAIN, CONTRACT, PINNED_EXP = 0, 1, 2


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.fixture(scope="module")
def ops(dev):
    import dispu_amd.tf_sampling as S
    import dispu_amd.tf_grouping as G
    import dispu_amd.tf_interpolate as I
    import dispu_amd.tf_nndistance as D
    import dispu_amd.tf_approxmatch as A
    import dispu_amd.nearest_neighbors as K
    from dispu_amd import _lib
    _lib.lib()  # fail loudly here if the HIP library is missing
    return dict(S=S, G=G, I=I, D=D, A=A, K=K)


def synth_patches(b, n, seed):
    from dispu_amd import synth
    return synth.patches(b, n, seed=seed)


# ---------------------------------------------------------------------------------- golden (reference) ----
def test_nn_distance_vs_reference_golden(ops, dev, golden_dir):
    z = g(golden_dir, "ref_nndistance.npz")
    d1, i1, d2, i2 = ops["D"].nn_distance(T(z["xyz1"], dev), T(z["xyz2"], dev), arith=PLAIN)
    assert np.array_equal(N(i1), z["idx1"]) and np.array_equal(N(i2), z["idx2"])
    assert np.array_equal(N(d1), z["dist1"]) and np.array_equal(N(d2), z["dist2"])


def test_interpolate_vs_reference_golden(ops, dev, golden_dir):
    z = g(golden_dir, "ref_interpolate.npz")
    d, i = ops["I"].three_nn(T(z["xyz1"], dev), T(z["xyz2"], dev))
    assert np.array_equal(N(i), z["idx"]) and np.array_equal(N(d), z["dist"])
    out = ops["I"].three_interpolate(T(z["points"], dev), T(z["idx"], dev), T(z["weight"], dev))
    assert np.array_equal(N(out), z["out"])
    gp = ops["I"].three_interpolate_grad(T(z["points"], dev), T(z["idx"], dev), T(z["weight"], dev), T(z["grad_out"], dev))
    assert np.allclose(N(gp), z["grad_points"], rtol=1e-5, atol=1e-5)      # atomic order differs


def test_grouping_vs_reference_golden(ops, dev, golden_dir):
    z = g(golden_dir, "ref_grouping.npz")
    idx, cnt = ops["G"].query_ball_point(float(z["radius"]), int(z["nsample"]), T(z["xyz1"], dev), T(z["xyz2"], dev), arith=PLAIN)
    assert np.array_equal(N(idx), z["idx"])
    out = ops["G"].group_point(T(z["points"], dev), T(z["idx"], dev))
    assert np.array_equal(N(out), z["out"])
    gp = ops["G"].group_point_grad(T(z["points"], dev), T(z["idx"], dev), T(z["grad_out"], dev))
    assert np.allclose(N(gp), z["grad_points"], rtol=1e-5, atol=1e-5)


def test_knn_vs_nanoflann_golden(ops, dev, golden_dir):
    z = g(golden_dir, "ref_knn.npz")
    k = int(z["k"])
    s, q = T(z["support"], dev), T(z["query"], dev)
    out = ops["K"].knn_batch(s, s, k, omp=True)
    assert out.dtype == torch.int64 and np.array_equal(N(out).astype(np.int32), z["idx_self"])
    assert np.array_equal(N(ops["K"].knn_batch(s, q, k)).astype(np.int32), z["idx_query"])
    assert np.array_equal(N(ops["K"].knn_query(k, s, q)), z["idx_query"])


def test_approxmatch_chunked_golden(ops, dev, golden_dir):
    """the frozen chunk-of-128 association (tests/golden/oracle_approxmatch_chunk128.npz): bit-exact in pinned-exp mode."""
    z = g(golden_dir, "oracle_approxmatch_chunk128.npz")
    m = ops["A"].approx_match(T(z["xyz1"], dev), T(z["xyz2"], dev), arith=CONTRACT | PINNED_EXP)
    assert np.array_equal(N(m), z["match_pinned"])
    assert np.allclose(N(ops["A"].match_cost(T(z["xyz1"], dev), T(z["xyz2"], dev), m)), z["cost"], rtol=1e-5)


def test_approxmatch_vs_golden(ops, dev, golden_dir):
    z = g(golden_dir, "ref_approxmatch.npz")
    x1, x2 = T(z["xyz1"], dev), T(z["xyz2"], dev)
    m = ops["A"].approx_match(x1, x2)
    mo = z["oracle_match"]
    assert N(m).shape == mo.shape
    # hardware v_exp_f32 vs libm expf: individual plan entries are ill-conditioned (min/max clamps in the
    # auction), the transported mass and the EMD are not -- entries 1e-3 abs, cost 1e-5 rel
    assert np.allclose(N(m), mo, atol=1e-3)
    assert np.allclose(N(m).sum(1), 1.0, atol=5e-5) and np.allclose(N(m).sum(2), 1.0, atol=5e-5)
    cost = ops["A"].match_cost(x1, x2, m)
    assert np.allclose(N(cost), z["ref_matchcost_on_oracle_match"], rtol=1e-5)    # north-star tolerance on EMD
    cost_o = ops["A"].match_cost(x1, x2, T(mo, dev))
    assert np.allclose(N(cost_o), z["ref_matchcost_on_oracle_match"], rtol=2e-6)
    g1, g2 = ops["A"].match_cost_grad(x1, x2, T(mo, dev))
    assert np.allclose(N(g1), z["ref_grad1_on_oracle_match"], atol=2e-5)
    assert np.allclose(N(g2), z["ref_grad2_on_oracle_match"], atol=2e-5)


def test_gpu_only_pins(ops, dev, golden_dir):
    z = g(golden_dir, "oracle_gpu_only.npz")
    f = T(z["fps_inp"], dev)
    assert np.array_equal(N(ops["S"].farthest_point_sample(96, f)), z["fps_idx_contract"])
    assert np.array_equal(N(ops["S"].farthest_point_sample(96, f, arith=PLAIN)), z["fps_idx_plain"])
    d, i = ops["G"].knn_point_2(17, T(z["feat"], dev), T(z["feat"], dev))
    assert np.array_equal(N(i)[..., 1], z["knn2_idx"]) and np.array_equal(N(i)[..., 0], np.broadcast_to(np.arange(2)[:, None, None], (2, 256, 17)))
    assert np.array_equal(N(d), z["knn2_dist"])


# ---------------------------------------------------------------------------------- oracle, seeded ----
@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 3, 3), (3, 64, 17), (2, 255, 64), (4, 256, 64), (2, 511, 100),
                                   (2, 512, 128), (2, 513, 128), (3, 1024, 384), (1, 2048, 24), (2, 3073, 40),
                                   (1, 5000, 33), (1, 9000, 20), (1, 20000, 12), (1, 24576, 10), (1, 30000, 8)])
@pytest.mark.parametrize("arith", [PLAIN, CONTRACT])
def test_fps_index_exact(ops, dev, b, n, m, arith):
    x = np.random.default_rng(n * 7 + m).random((b, n, 3)).astype(np.float32)
    got = N(ops["S"].farthest_point_sample(m, T(x, dev), arith=arith))
    assert np.array_equal(got, O.farthest_point_sample(m, x, contract=arith))


@pytest.mark.parametrize("b,n,m,kind", [(1, 24576, 8192, "surface"), (2, 24576, 700, "cube"), (2, 12000, 3000, "surface"), (3, 5000, 1200, "cube"),
                                        (1, 8193, 64, "cube"), (2, 6000, 600, "dups"), (1, 4097, 4097, "grid"),
                                        # the wave-skipping kernel (csrc/fps_wave.hip: 4096 < n <= 24576, m >= 64): ties inside a
                                        # lane, inside a wave and across waves; ragged last wave; both register layouts
                                        (1, 16385, 300, "dups"), (1, 10000, 800, "grid"), (2, 24576, 1500, "dups"), (1, 20000, 900, "grid"),
                                        (3, 9001, 257, "surface"), (1, 16384, 2048, "cube")])
def test_fps_large_clouds_index_exact(ops, dev, b, n, m, kind):
    """Large clouds at the whole-cloud test path's shape (1, 24576, 8192) (DisPU/model.py:375) and around the register-kernel
    boundaries: the sampled indices equal the oracle's exactly, including tie decisions (duplicated points, grids) --
    tf_sampling_g.cu:105-170."""
    rng = np.random.default_rng(n + m)
    if kind == "surface":                                  # points on a sphere-like 2-manifold, like merged patches
        g = rng.standard_normal((b, n, 3))
        x = (g / np.linalg.norm(g, axis=2, keepdims=True) * (1 + 0.02 * rng.standard_normal((b, n, 1)))).astype(np.float32)
    elif kind == "cube":
        x = rng.random((b, n, 3)).astype(np.float32)
    elif kind == "dups":
        x = np.repeat(rng.random((b, n // 4, 3)).astype(np.float32), 4, axis=1)
        x = x[:, rng.permutation(x.shape[1])]
    else:
        side = int(np.ceil(n ** (1 / 3)))
        gx = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3)[:n]
        x = np.broadcast_to(gx.astype(np.float32)[None], (b, n, 3)).copy()
    n = x.shape[1]
    for arith in (CONTRACT, PLAIN):
        got = N(ops["S"].farthest_point_sample(m, T(x, dev), arith=arith))
        want = O.farthest_point_sample(m, x, contract=arith)
        assert np.array_equal(got, want), (kind, arith, int(np.argmax((got != want).any(0))))
        if m > 2000:
            break                                          # one flavour is enough at the big shapes (the oracle takes seconds)


def test_fps_region_kernels_random_shapes(ops, dev):
    """Seeded random shapes over the whole range of the region-skipping kernels (4096 < n <= 24576, both register layouts, ragged
    last wave / region) on clouds built to provoke ties: points duplicated 2 - 6 times, points snapped to a coarse lattice (many
    equal distances across regions and waves), thin slabs -- indices equal to the oracle's."""
    rng = np.random.default_rng(20260928)
    for case in range(14):
        n = int(rng.integers(4097, 24577))
        m = int(rng.integers(64, 360))
        b = int(rng.integers(1, 4))
        kind = case % 4
        if kind == 0:
            x = rng.random((b, n, 3))
        elif kind == 1:                                   # duplicates
            rep = int(rng.integers(2, 7))
            base = rng.random((b, (n + rep - 1) // rep, 3))
            x = np.repeat(base, rep, axis=1)[:, :n]
            x = x[:, rng.permutation(n)]
        elif kind == 2:                                   # lattice: exact ties everywhere
            q = int(rng.integers(6, 30))
            x = np.round(rng.random((b, n, 3)) * q) / q
        else:                                             # a thin slab
            x = rng.random((b, n, 3)) * np.array([1.0, 1.0, 0.01])
        x = x.astype(np.float32)
        arith = CONTRACT if case % 2 == 0 else PLAIN
        got = N(ops["S"].farthest_point_sample(m, T(x, dev), arith=arith))
        want = O.farthest_point_sample(m, x, contract=arith)
        assert np.array_equal(got, want), (case, b, n, m, kind, int(np.argmax((got != want).any(0))))


def test_fps_abi_with_and_without_scratch(dev):
    """dispu_fps through the C ABI at a wave-skipping size: with the scratch dispu_fps_scratch_bytes asks for (the permutation of
    csrc/fps_wave.hip) and without any (NULL: the dense register kernel answers) -- same indices, equal to the oracle's."""
    from dispu_amd import _lib
    L = _lib.lib()
    b, n, m = 2, 10000, 300
    x = np.random.default_rng(77).random((b, n, 3)).astype(np.float32)
    tx = T(x, dev)
    nbytes = L.dispu_fps_scratch_bytes(b, n, m)
    assert nbytes == b * n * 4
    outs = []
    for scratch in (torch.empty(nbytes // 4, dtype=torch.float32, device=dev), None):
        out = torch.full((b, m), -1, dtype=torch.int32, device=dev)
        _lib.check(L.dispu_fps(b, n, m, tx.data_ptr(), scratch.data_ptr() if scratch is not None else None, out.data_ptr(),
                               CONTRACT, _lib.stream_ptr(dev)), "dispu_fps")
        outs.append(N(out))
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], O.farthest_point_sample(m, x, contract=CONTRACT))


def test_fps_ties_and_duplicates(ops, dev):
    """Adversarial: grids (many exact ties), duplicated points, more samples than distinct points."""
    gx = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(7), indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    for m in (5, 64, 700):
        assert np.array_equal(N(ops["S"].farthest_point_sample(m, T(gx, dev))), O.farthest_point_sample(m, gx))
    dup = np.repeat(np.random.default_rng(0).random((2, 150, 3)).astype(np.float32), 5, axis=1)   # n = 750
    assert np.array_equal(N(ops["S"].farthest_point_sample(200, T(dup, dev))), O.farthest_point_sample(200, dup))
    big = np.zeros((1, 1300, 3), np.float32)
    big[0, [3, 515, 1027, 600], 1] = 2.0
    assert np.array_equal(N(ops["S"].farthest_point_sample(6, T(big, dev))), O.farthest_point_sample(6, big))


def test_gather_point_and_grad(ops, dev):
    rng = np.random.default_rng(5)
    inp = rng.standard_normal((3, 300, 3)).astype(np.float32)
    idx = rng.integers(0, 300, (3, 1000)).astype(np.int32)
    assert np.array_equal(N(ops["S"].gather_point(T(inp, dev), T(idx, dev))), O.gather_point(inp, idx))
    og = rng.standard_normal((3, 1000, 3)).astype(np.float32)
    got = N(ops["S"].gather_point_grad(T(inp, dev), T(idx, dev), T(og, dev)))
    assert np.allclose(got, O.gather_point_grad(inp, idx, og), rtol=1e-5, atol=1e-5)
    # autograd wiring == registered gradient (tf_sampling.py:43-47)
    ti = T(inp, dev).requires_grad_(True)
    ops["S"].gather_point(ti, T(idx, dev)).backward(T(og, dev))
    assert np.allclose(N(ti.grad), got, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("arith", [PLAIN, CONTRACT])
@pytest.mark.parametrize("b,n,m,r,ns", [(2, 1024, 1024, 0.07, 20), (3, 1500, 300, 0.2, 32), (1, 5, 7, 0.5, 4), (2, 2500, 64, 0.1, 64),
                                        (2, 65, 65, 0.3, 70), (3, 128, 77, 0.25, 8), (2, 300, 300, 0.2, 33), (1, 512, 100, 0.5, 64),
                                        (2, 1000, 1023, 0.15, 96), (8, 1024, 1024, 0.07, 20), (2, 4096, 4096, 0.07, 20), (1, 8200, 500, 0.05, 40),
                                        (2, 1025, 1025, 2.0, 200), (1, 3000, 100, 1e-4, 8)])
def test_query_ball_index_exact(ops, dev, b, n, m, r, ns, arith):
    x = synth_patches(b, n, seed=n)
    q = x[:, :m] if m <= n else np.concatenate([x, x[:, : m - n]], 1)
    idx, cnt = ops["G"].query_ball_point(r, ns, T(x, dev), T(q, dev), arith=arith)
    oi, oc = O.query_ball_point(r, ns, x, q, contract=arith)
    assert np.array_equal(N(cnt), oc) and np.array_equal(N(idx), oi)


@pytest.mark.parametrize("n,r", [(1024, 0.25), (3000, 0.25), (512, 1e-3), (1024, 3.0)])
def test_query_ball_candidates_on_the_radius(ops, dev, n, r):
    """The wave kernels decide `max(sqrtf(d2), 1e-20) < radius` from d2 alone outside a 4e-6 band around radius^2 and take the exact
    square root inside it (csrc/grouping.hip:qb_hit): candidates placed within a few ulps of the radius, on both sides and exactly on it,
    must be classified as the reference's expression classifies them (the oracle evaluates it literally); both distance flavours."""
    rng = np.random.default_rng(n)
    q = rng.random((2, 40, 3)).astype(np.float32)
    x = rng.random((2, n, 3)).astype(np.float32) * 4.0 + 10.0            # far away: never hit
    u = rng.standard_normal((2, 40, 12, 3))
    u /= np.linalg.norm(u, axis=-1, keepdims=True)
    scale = np.float64(r) * (1.0 + np.array([-3e-6, -1e-6, -3e-7, -1e-7, -3e-8, 0.0, 0.0, 3e-8, 1e-7, 3e-7, 1e-6, 3e-6]))
    shell = (q[:, :, None, :].astype(np.float64) + u * scale[None, None, :, None]).astype(np.float32).reshape(2, 480, 3)
    pos = rng.permutation(n)[:480]
    x[:, pos] = shell
    for contract in (0, 1):
        idx, cnt = ops["G"].query_ball_point(r, 16, T(x, dev), T(q, dev), arith=contract)
        oi, oc = O.query_ball_point(r, 16, x, q, contract=contract)
        assert np.array_equal(N(cnt), oc) and np.array_equal(N(idx), oi)


def test_query_ball_no_hit_rows_and_radius_tensor(ops, dev):
    x = np.random.default_rng(1).random((2, 50, 3)).astype(np.float32)
    q = x[:, :9] + 100.0
    idx, cnt = ops["G"].query_ball_point(torch.tensor([0.3, 9999.0], device=dev), 5, T(x, dev), T(q, dev))
    assert (N(cnt) == 0).all() and (N(idx) == 0).all()      # only radius[0] is read: cloud 1 also finds nothing


@pytest.mark.parametrize("c", [1, 3, 4, 7, 16, 64, 128, 134])
def test_group_point_bit_exact(ops, dev, c):
    rng = np.random.default_rng(c)
    pts = rng.standard_normal((3, 257, c)).astype(np.float32)
    idx = rng.integers(0, 257, (3, 100, 16)).astype(np.int32)
    assert np.array_equal(N(ops["G"].group_point(T(pts, dev), T(idx, dev))), O.group_point(pts, idx))
    go = rng.standard_normal((3, 100, 16, c)).astype(np.float32)
    assert np.allclose(N(ops["G"].group_point_grad(T(pts, dev), T(idx, dev), T(go, dev))), O.group_point_grad(pts, idx, go),
                       rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("b,n,m,k", [(2, 1024, 1024, 16), (3, 300, 77, 8), (1, 40, 40, 32), (2, 2500, 130, 20), (1, 5, 5, 5), (2, 64, 64, 1)])
def test_knn_xyz_index_exact(ops, dev, b, n, m, k):
    s = synth_patches(b, n, seed=k)
    q = s[:, :m] if m <= n else synth_patches(b, m, seed=k + 1)
    idx, dist = ops["K"].knn_batch(T(s, dev), T(q, dev), k, return_dist=True)
    oi, od = O.knn_batch(s, q, k, return_dist=True)
    assert np.array_equal(N(idx), oi) and np.array_equal(N(dist), od)


def test_knn_xyz_ties_lower_index_first(ops, dev):
    s = np.zeros((1, 100, 3), np.float32)
    s[0, 50:, 0] = 1.0
    idx = N(ops["K"].knn_batch(T(s, dev), T(s[:, :1], dev), 8))
    assert idx[0, 0].tolist() == list(range(8))


@pytest.mark.parametrize("c,k", [(3, 5), (3, 16), (24, 17), (48, 17), (64, 20), (5, 3), (128, 32)])
def test_knn_point_variants(ops, dev, c, k):
    rng = np.random.default_rng(c * 31 + k)
    a = rng.standard_normal((2, 256, c)).astype(np.float32)
    q = rng.standard_normal((2, 90, c)).astype(np.float32)
    val, idx = ops["G"].knn_point(k, T(a, dev), T(q, dev))
    ov, oi = O.knn_point(k, a, q)
    assert np.array_equal(N(idx), oi) and np.array_equal(N(val), ov)
    assert (N(val) <= 0).all()                                            # NEGATIVE squared distances
    d2, i2 = ops["G"].knn_point_2(k, T(a, dev), T(q, dev))
    od, oi2 = O.knn_point_2(k, a, q)
    assert np.array_equal(N(i2), oi2) and np.array_equal(N(d2), od)


@pytest.mark.parametrize("n,m,c,k", [(17, 5, 24, 9), (64, 64, 48, 17), (65, 37, 24, 9), (100, 100, 12, 17), (200, 33, 48, 20),
                                     (255, 255, 7, 17), (256, 300, 64, 32), (300, 40, 24, 9), (500, 77, 48, 17), (700, 20, 16, 5)])
def test_knn_point_2_cloud_sizes(ops, dev, n, m, c, k):
    """Feature-space k-NN across the kernel variants: n <= 256 (MFMA dot products, candidate tiles of 16, partial last
    tile), n <= 512 (VALU dots, R = 8) and larger (lane-per-query); queries not a multiple of the 16 per workgroup."""
    rng = np.random.default_rng(n * 7 + m)
    a = rng.standard_normal((3, n, c)).astype(np.float32)
    q = np.concatenate([a[:, : min(m, n) // 2], rng.standard_normal((3, m - min(m, n) // 2, c)).astype(np.float32)], 1)
    d2, i2 = ops["G"].knn_point_2(k, T(a, dev), T(q, dev))
    od, oi2 = O.knn_point_2(k, a, q)
    assert np.array_equal(N(i2), oi2) and np.array_equal(N(d2), od)


@pytest.mark.parametrize("arith", [PLAIN, CONTRACT])
@pytest.mark.parametrize("b,n,m", [(2, 1024, 384), (3, 384, 128), (2, 128, 1), (1, 2, 2), (2, 1500, 2500)])
def test_three_nn_exact(ops, dev, b, n, m, arith):
    rng = np.random.default_rng(n * 3 + m)
    x1, x2 = rng.random((b, n, 3)).astype(np.float32), rng.random((b, m, 3)).astype(np.float32)
    d, i = ops["I"].three_nn(T(x1, dev), T(x2, dev), arith=arith)
    od, oi = O.three_nn(x1, x2, contract=arith)
    assert np.array_equal(N(i), oi) and np.array_equal(N(d), od)


@pytest.mark.parametrize("arith", [PLAIN, CONTRACT])
@pytest.mark.parametrize("b,n,m", [(4, 1024, 1024), (2, 4096, 4096), (3, 100, 777), (1, 1, 1), (2, 2049, 513),
                                   # either side of the reference kernel's 512-point staging batch (tf_nndistance_g.cu:6) and of 6 x 512
                                   (2, 511, 513), (2, 512, 512), (1, 513, 511), (1, 3072, 3073), (1, 3073, 3072), (2, 512, 3073)])
def test_nn_distance_exact(ops, dev, b, n, m, arith):
    rng = np.random.default_rng(n + m)
    x1, x2 = rng.standard_normal((b, n, 3)).astype(np.float32), rng.standard_normal((b, m, 3)).astype(np.float32)
    d1, i1, d2, i2 = ops["D"].nn_distance(T(x1, dev), T(x2, dev), arith=arith)
    o = O.nn_distance(x1, x2, contract=arith)
    for got, want in zip((d1, i1, d2, i2), o):
        assert np.array_equal(N(got), want)


def test_nn_distance_grad_and_autograd(ops, dev):
    x1, x2 = synth_patches(2, 300, seed=1), synth_patches(2, 200, seed=2)
    rng = np.random.default_rng(0)
    gd1, gd2 = rng.standard_normal((2, 300)).astype(np.float32), rng.standard_normal((2, 200)).astype(np.float32)
    _, i1, _, i2 = O.nn_distance(x1, x2)
    want = O.nn_distance_grad(x1, x2, gd1, i1, gd2, i2)
    got = ops["D"].nn_distance_grad(T(x1, dev), T(x2, dev), T(gd1, dev), T(i1, dev), T(gd2, dev), T(i2, dev))
    assert np.allclose(N(got[0]), want[0], atol=1e-5) and np.allclose(N(got[1]), want[1], atol=1e-5)
    t1, t2 = T(x1, dev).requires_grad_(True), T(x2, dev).requires_grad_(True)
    d1, _, d2, _ = ops["D"].nn_distance(t1, t2)
    ((d1 * T(gd1, dev)).sum() + (d2 * T(gd2, dev)).sum()).backward()
    assert np.allclose(N(t1.grad), want[0], atol=1e-5) and np.allclose(N(t2.grad), want[1], atol=1e-5)


@pytest.mark.parametrize("b,n,m", [(2, 256, 256), (1, 1024, 1024), (2, 100, 300), (2, 300, 100), (1, 1100, 1030), (3, 1, 5), (2, 129, 127),
                                   (1, 2048, 512),
                                   # either side of the reference kernel's 1024-point batch (tf_approxmatch_g.cu:11 `Block = 1024`)
                                   (1, 1023, 1025), (1, 1025, 1023), (1, 1024, 1023)])
def test_approx_match_and_cost(ops, dev, b, n, m):
    if min(n, m) < 8:                 # the patch synthesiser normalises by the max radius: undefined for a single point
        rng = np.random.default_rng(n * 1000 + m)
        x1, x2 = rng.random((b, n, 3), dtype=np.float32), rng.random((b, m, 3), dtype=np.float32)
    else:
        x1, x2 = synth_patches(b, n, seed=n), synth_patches(b, m, seed=m + 1)
    # parity mode: pinned exp on both sides and the kernels' summation order (partial sums per 128 partners, added in
    # ascending order: oracle chunk = AM_CHUNK) -> the whole auction is bit-reproducible
    mp = ops["A"].approx_match(T(x1, dev), T(x2, dev), arith=CONTRACT | PINNED_EXP)
    assert np.array_equal(N(mp), O.approx_match(x1, x2, contract=1, pinned_exp=True, chunk=O.AM_CHUNK))
    mp0 = ops["A"].approx_match(T(x1, dev), T(x2, dev), arith=PLAIN | PINNED_EXP)
    assert np.array_equal(N(mp0), O.approx_match(x1, x2, contract=0, pinned_exp=True, chunk=O.AM_CHUNK))
    # production mode: hardware exp (like the reference's __expf) vs libm expf in the oracle
    match = ops["A"].approx_match(T(x1, dev), T(x2, dev))
    mo = O.approx_match(x1, x2)
    dm = np.abs(N(match) - mo)
    # single plan entries are ill-conditioned w.r.t. 1-ulp exp differences (min/max clamps of the auction,
    # worst when n != m); the EMD itself is not: entries 1e-3 abs for all but <1e-5 of them, cost 1e-5 rel.
    assert dm.max() < 5e-2 and np.mean(dm > 1e-3) < 1e-5
    cost = N(ops["A"].match_cost(T(x1, dev), T(x2, dev), match))
    co = O.match_cost(x1, x2, mo)
    assert np.allclose(cost, co, rtol=1e-5)                                # <= 1e-5 on EMD (north star)
    g1, g2 = ops["A"].match_cost_grad(T(x1, dev), T(x2, dev), T(mo, dev))
    o1, o2 = O.match_cost_grad(x1, x2, mo)
    assert np.allclose(N(g1), o1, atol=3e-5) and np.allclose(N(g2), o2, atol=3e-5)


def test_match_cost_autograd_scaling(ops, dev):
    x1, x2 = synth_patches(2, 128, seed=9), synth_patches(2, 128, seed=10)
    t1, t2 = T(x1, dev).requires_grad_(True), T(x2, dev).requires_grad_(True)
    match = ops["A"].approx_match(t1.detach(), t2.detach())
    w = torch.tensor([2.0, -3.0], device=dev)
    (ops["A"].match_cost(t1, t2, match) * w).sum().backward()
    g1, g2 = ops["A"].match_cost_grad(t1.detach(), t2.detach(), match)
    assert torch.allclose(t1.grad, g1 * w.view(2, 1, 1)) and torch.allclose(t2.grad, g2 * w.view(2, 1, 1))


def test_gradient_checks_of_the_reference_tests(ops, dev):
    """tf_grouping_op_test.py:9-25 and tf_interpolate_op_test.py:9-22 restated: the analytic gradient of
    group_point / three_interpolate w.r.t. `points` equals a finite-difference gradient (< 1e-4... the ops are
    linear in `points`, so the check is exact up to fp32 rounding)."""
    rng = np.random.default_rng(0)
    pts = rng.random((1, 128, 16)).astype(np.float32)
    xyz1, xyz2 = rng.random((1, 128, 3)).astype(np.float32), rng.random((1, 8, 3)).astype(np.float32)
    idx, _ = ops["G"].query_ball_point(0.3, 32, T(xyz1, dev), T(xyz2, dev))
    tp = T(pts, dev).requires_grad_(True)
    w = torch.randn(1, 8, 32, 16, device=dev)
    (ops["G"].group_point(tp, idx) * w).sum().backward()
    eps = 1e-2
    for (i, c) in [(0, 0), (17, 3), (127, 15)]:
        e = torch.zeros_like(tp); e[0, i, c] = eps
        num = ((ops["G"].group_point(tp.detach() + e, idx) - ops["G"].group_point(tp.detach() - e, idx)) * w).sum() / (2 * eps)
        assert abs(float(num) - float(tp.grad[0, i, c])) < 1e-3
    pts2 = rng.random((1, 8, 16)).astype(np.float32)
    idx3 = T(rng.integers(0, 8, (1, 128, 3)).astype(np.int32), dev)
    w3 = torch.full((1, 128, 3), 1.0 / 3.0, device=dev)
    tp2 = T(pts2, dev).requires_grad_(True)
    wo = torch.randn(1, 128, 16, device=dev)
    (ops["I"].three_interpolate(tp2, idx3, w3) * wo).sum().backward()
    for (i, c) in [(0, 0), (5, 7)]:
        e = torch.zeros_like(tp2); e[0, i, c] = eps
        num = ((ops["I"].three_interpolate(tp2.detach() + e, idx3, w3) - ops["I"].three_interpolate(tp2.detach() - e, idx3, w3)) * wo).sum() / (2 * eps)
        assert abs(float(num) - float(tp2.grad[0, i, c])) < 1e-3


# ---------------------------------------------------------------------------------- full-size properties ----
def test_full_size_properties(ops, dev):
    """BASELINE sizes where the CPU oracle would take too long: size-independent properties."""
    b, n = 32, 1024
    x = T(synth_patches(b, n, seed=77), dev)
    # k-NN: self first, distances ascending, idx consistent with distances, symmetric count sanity
    idx, dist = ops["K"].knn_batch(x, x, 16, return_dist=True)
    assert (idx[:, :, 0] == torch.arange(n, device=dev)).all() and (dist[:, :, 0] == 0).all()
    assert (dist[:, :, 1:] >= dist[:, :, :-1]).all()
    nb = torch.gather(x.unsqueeze(1).expand(b, n, n, 3), 2, idx.unsqueeze(-1).expand(b, n, 16, 3))
    d_chk = ((nb - x.unsqueeze(2)) ** 2).sum(-1)
    assert torch.allclose(d_chk, dist, atol=1e-6)
    # Chamfer of a cloud with itself is zero with identity matches
    d1, i1, d2, i2 = ops["D"].nn_distance(x, x)
    assert (d1 == 0).all() and (i1 == torch.arange(n, device=dev)).all() and (i2 == i1).all()
    # FPS of 8 x 24576 points (test-time shape): a permutation prefix, first index 0
    big = torch.rand(2, 24576, 3, device=dev)
    fi = ops["S"].farthest_point_sample(2048, big)
    assert (fi[:, 0] == 0).all() and all(len(set(r.tolist())) == 2048 for r in fi)
    # gather(idx) round trip: gather_point(FPS idx) == advanced indexing
    assert torch.equal(ops["S"].gather_point(big, fi), torch.gather(big, 1, fi.long().unsqueeze(-1).expand(-1, -1, 3)))
    # EMD: match rows/cols sum to one at 1024^2; cost(x, x) is ~0 relative to cost(x, y)
    y = T(synth_patches(4, n, seed=78), dev)
    m = ops["A"].approx_match(x[:4], y)
    assert torch.allclose(m.sum(1), torch.ones(4, n, device=dev), atol=1e-4)
    assert torch.allclose(m.sum(2), torch.ones(4, n, device=dev), atol=1e-4)
    c_xy = ops["A"].match_cost(x[:4], y, m)
    c_xx = ops["A"].match_cost(x[:4], x[:4], ops["A"].approx_match(x[:4], x[:4]))
    assert (c_xx < 0.05 * c_xy).all()


@pytest.mark.parametrize("b,n,m,k", [(4, 1024, 1024, 16), (2, 700, 333, 20), (3, 65, 65, 32), (2, 1000, 50, 1)])
def test_knn_xyz_wave_path_equals_lane_path(ops, dev, b, n, m, k):
    """dispu_knn_xyz has two formulations (wave-per-query for n <= 1024, lane-per-query otherwise): identical output."""
    from dispu_amd import _lib
    rng = np.random.default_rng(n + k)
    s = rng.random((b, n, 3)).astype(np.float32)
    s[:, 5] = s[:, 17]                                        # exact duplicates -> ties must resolve to the lower index
    q = np.concatenate([s[:, : m // 2], rng.random((b, m - m // 2, 3)).astype(np.float32)], 1)
    ts, tq = T(s, dev), T(q, dev)
    res = []
    for arith in (PLAIN, PLAIN | 4, CONTRACT, CONTRACT | 4):
        i, d = ops["K"].knn_batch(ts, tq, k, return_dist=True, arith=arith)
        res.append((N(i), N(d)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[2][0], res[3][0]) and np.array_equal(res[2][1], res[3][1])
    oi, od = O.knn_batch(s, q, k, return_dist=True)
    assert np.array_equal(res[0][0], oi) and np.array_equal(res[0][1], od)


@pytest.mark.parametrize("n,ndup,k", [(1024, 40, 16), (1024, 100, 16), (1024, 300, 16), (600, 90, 32), (1024, 1024, 8)])
def test_knn_xyz_prefilter_degenerate_clouds(ops, dev, n, ndup, k):
    """The n > 256 wave kernel prefilters with a threshold and ranks the survivors (<= 64, <= 128) or falls back to the
    full sort (> 128): clouds with `ndup` coincident points force each of the three paths; ties -> lower index."""
    rng = np.random.default_rng(n + ndup)
    s = rng.random((2, n, 3)).astype(np.float32)
    where = rng.permutation(n)[:ndup]
    s[:, where] = s[:, where[:1]]                                # ndup copies of one point
    s[1, ::4] = s[1, 1::4] = s[1, 2::4] + np.float32(1e-3)      # and near-coincident groups 1 apart in memory
    q = np.concatenate([s[:, where[:3]], s[:, :40], rng.random((2, 21, 3)).astype(np.float32)], 1)
    i, d = ops["K"].knn_batch(T(s, dev), T(q, dev), k, return_dist=True)
    oi, od = O.knn_batch(s, q, k, return_dist=True)
    assert np.array_equal(N(i), oi) and np.array_equal(N(d), od)
    # children of one parent 256 apart in memory (the generator's coarse clouds)
    par = rng.random((2, 256, 3)).astype(np.float32)
    c = (par[:, None] + 0.01 * rng.standard_normal((2, 4, 256, 3)).astype(np.float32)).reshape(2, 1024, 3)
    i, d = ops["K"].knn_batch(T(c, dev), T(c, dev), 16, return_dist=True)
    oi, od = O.knn_batch(c, c, 16, return_dist=True)
    assert np.array_equal(N(i), oi) and np.array_equal(N(d), od)


# ---- shapes beyond the register-resident fast paths: general radix-select kernel (csrc/knn_general.hip) ---------------
@pytest.mark.parametrize("b,n,m,k", [(2, 1024, 200, 64), (1, 3000, 77, 33), (2, 500, 500, 500), (1, 24576, 40, 256), (1, 5000, 9, 1000)])
def test_knn_xyz_large_k(ops, dev, b, n, m, k):
    """nanoflann takes any K (libs/nearest_neighbors/knn_.cxx:104-135)."""
    s = synth_patches(b, n, seed=n + k)
    q = s[:, :m]
    for arith, contract in ((PLAIN, 0), (CONTRACT, 1)):
        idx, dist = ops["K"].knn_batch(T(s, dev), T(q, dev), k, return_dist=True, arith=arith)
        oi, od = O.knn_batch(s, q, k, contract=contract, return_dist=True)
        assert np.array_equal(N(idx), oi) and np.array_equal(N(dist), od)


def test_knn_general_path_agrees_with_fast_path(ops, dev):
    """k = 33 takes the general kernel, k = 32 the register-resident one: the first 32 columns must be identical; duplicate
    points make the threshold a tie (index-ordered quota of the radix select)."""
    s = synth_patches(2, 700, seed=3)
    s[:, 100:160] = s[:, 5:6]                                    # 61 copies of one point
    ts = T(s, dev)
    i33, d33 = ops["K"].knn_batch(ts, ts, 33, return_dist=True)
    i32, d32 = ops["K"].knn_batch(ts, ts, 32, return_dist=True)
    assert np.array_equal(N(i33)[..., :32], N(i32)) and np.array_equal(N(d33)[..., :32], N(d32))
    oi, od = O.knn_batch(s, s, 33, return_dist=True)
    assert np.array_equal(N(i33), oi) and np.array_equal(N(d33), od)


@pytest.mark.parametrize("n,m,c,k", [(300, 50, 256, 17), (256, 64, 200, 40), (1024, 33, 24, 48), (90, 90, 131, 90), (400, 10, 3, 100)])
def test_knn_point_large_k_and_c(ops, dev, n, m, c, k):
    """knn_point / knn_point_2 are tf.nn.top_k over a full distance matrix: any k, any channel count
    (tf_ops/grouping/tf_grouping.py:95-141)."""
    rng = np.random.default_rng(n + 13 * c + k)
    a = rng.standard_normal((2, n, c)).astype(np.float32)
    q = np.concatenate([a[:, : m // 2], rng.standard_normal((2, m - m // 2, c)).astype(np.float32)], 1)
    val, idx = ops["G"].knn_point(k, T(a, dev), T(q, dev))
    ov, oi = O.knn_point(k, a, q)
    assert np.array_equal(N(idx), oi) and np.array_equal(N(val), ov)
    d2, i2 = ops["G"].knn_point_2(k, T(a, dev), T(q, dev))
    od, oi2 = O.knn_point_2(k, a, q)
    assert np.array_equal(N(i2), oi2) and np.array_equal(N(d2), od)


@pytest.mark.parametrize("b,n,m,k", [(2, 4096, 700, 16), (1, 3000, 500, 8), (1, 1025, 64, 32), (2, 8192, 100, 16), (1, 5000, 5000, 17), (3, 2048, 2048, 1)])
def test_knn_xyz_chunked_path(ops, dev, b, n, m, k):
    """Clouds of 1025 .. 8192 points (the second generator pass of 16x upsampling queries 4096-point clouds): per-chunk
    wave kernel + merge (csrc/knn_wave.hip) == one scan over the whole cloud == the lane-per-query kernel, incl. duplicates."""
    s = synth_patches(b, n, seed=n + k)
    s[:, 1500 % n] = s[:, 7]                                   # equal distances in different chunks -> the lower index first
    s[:, n - 1] = s[:, 7]
    q = s[:, :m]
    ts, tq = T(s, dev), T(q, dev)
    oi, od = O.knn_batch(s, q, k, return_dist=True)
    i, d = ops["K"].knn_batch(ts, tq, k, return_dist=True)
    assert np.array_equal(N(i), oi) and np.array_equal(N(d), od)
    i2, d2 = ops["K"].knn_batch(ts, tq, k, return_dist=True, arith=PLAIN | 4)       # DISPU_KNN_LANE_PER_QUERY
    assert np.array_equal(N(i2), oi) and np.array_equal(N(d2), od)
    oi1, od1 = O.knn_batch(s, q, k, contract=1, return_dist=True)
    i3, d3 = ops["K"].knn_batch(ts, tq, k, return_dist=True, arith=CONTRACT)
    assert np.array_equal(N(i3), oi1) and np.array_equal(N(d3), od1)


@pytest.mark.parametrize("n,ndup,k", [(4096, 40, 16), (3000, 100, 16), (2048, 300, 16), (1500, 90, 32), (4096, 4096, 8)])
def test_knn_xyz_lds_path_degenerate_clouds(ops, dev, n, ndup, k):
    """1024 < n <= 4096 takes the single-pass kernel with the cloud in LDS (csrc/knn_wave.hip:knn_xyz_lds_kernel): `ndup` coincident
    points force its three selection paths (<= 64 survivors, <= 128, the k-round arg-min beyond); ties -> lower index; both distance
    flavours; children of one parent 1024 apart in memory (the second 16x pass's coarse clouds)."""
    rng = np.random.default_rng(n + ndup)
    s = rng.random((2, n, 3)).astype(np.float32)
    where = rng.permutation(n)[:ndup]
    s[:, where] = s[:, where[:1]]
    q = np.concatenate([s[:, where[:3]], s[:, :40], rng.random((2, 21, 3)).astype(np.float32)], 1)
    for contract, arith in ((0, PLAIN), (1, CONTRACT)):
        i, d = ops["K"].knn_batch(T(s, dev), T(q, dev), k, return_dist=True, arith=arith)
        oi, od = O.knn_batch(s, q, k, contract=contract, return_dist=True)
        assert np.array_equal(N(i), oi) and np.array_equal(N(d), od)
    par = rng.random((1, 1024, 3)).astype(np.float32)
    c = (par[:, None] + 0.01 * rng.standard_normal((1, 4, 1024, 3)).astype(np.float32)).reshape(1, 4096, 3)
    i, d = ops["K"].knn_batch(T(c, dev), T(c[:, ::8], dev), 16, return_dist=True)
    oi, od = O.knn_batch(c, c[:, ::8], 16, return_dist=True)
    assert np.array_equal(N(i), oi) and np.array_equal(N(d), od)


def test_empty_and_ragged_inputs(ops, dev):
    """Edge cases the reference's shape checks admit: empty batches / query sets, single points, k == n, row counts that are not
    a multiple of any tile (the reference tests none of these explicitly; its kernels loop `for (i = blockIdx.x; i < b; ...)`
    and simply do nothing for b == 0)."""
    S, G, I, D, A, K = (ops[k] for k in "SGIDAK")
    z3 = lambda *s: torch.zeros(s, device=dev)
    # empty batch
    assert tuple(S.farthest_point_sample(4, z3(0, 10, 3)).shape) == (0, 4)
    assert tuple(S.gather_point(z3(0, 10, 3), torch.zeros((0, 5), dtype=torch.int32, device=dev)).shape) == (0, 5, 3)
    assert tuple(G.group_point(z3(0, 10, 8), torch.zeros((0, 5, 4), dtype=torch.int32, device=dev)).shape) == (0, 5, 4, 8)
    assert tuple(D.nn_distance(z3(0, 10, 3), z3(0, 7, 3))[0].shape) == (0, 10)
    assert tuple(A.approx_match(z3(0, 10, 3), z3(0, 7, 3)).shape) == (0, 7, 10)
    assert tuple(A.match_cost(z3(0, 10, 3), z3(0, 7, 3), z3(0, 7, 10)).shape) == (0,)
    assert tuple(K.knn_batch(z3(0, 10, 3), z3(0, 4, 3), 3).shape) == (0, 4, 3)
    # empty query / sample sets
    x = torch.rand(2, 33, 3, device=dev)
    assert tuple(G.query_ball_point(0.5, 4, x, z3(2, 0, 3))[0].shape) == (2, 0, 4)
    assert tuple(G.group_point(x, torch.zeros((2, 0, 4), dtype=torch.int32, device=dev)).shape) == (2, 0, 4, 3)
    assert tuple(K.knn_batch(x, z3(2, 0, 3), 5).shape) == (2, 0, 5)
    # single point, k == n, odd sizes through every row-slot kernel
    one = torch.rand(3, 1, 3, device=dev)
    assert N(S.farthest_point_sample(1, one)).tolist() == [[0], [0], [0]]
    assert N(K.knn_batch(one, one, 1)).reshape(-1).tolist() == [0, 0, 0]
    xn = np.random.default_rng(2).random((2, 37, 3)).astype(np.float32)
    assert np.array_equal(N(K.knn_batch(T(xn, dev), T(xn, dev), 37)), O.knn_batch(xn, xn, 37))
    rng = np.random.default_rng(4)
    for c in (1, 2, 3, 5, 12, 130, 260):
        pts = rng.standard_normal((3, 19, c)).astype(np.float32)
        idx = rng.integers(0, 19, (3, 7, 5)).astype(np.int32)
        assert np.array_equal(N(G.group_point(T(pts, dev), T(idx, dev))), O.group_point(pts, idx))
        i3 = rng.integers(0, 19, (3, 11, 3)).astype(np.int32)
        w3 = rng.random((3, 11, 3)).astype(np.float32)
        assert np.array_equal(N(I.three_interpolate(T(pts, dev), T(i3, dev), T(w3, dev))), O.three_interpolate(pts, i3, w3))
    idx1 = rng.integers(0, 37, (2, 1031)).astype(np.int32)              # gather_xyz: a ragged last pass (1031 = 1024 + 7)
    assert np.array_equal(N(S.gather_point(T(xn, dev), T(idx1, dev))), O.gather_point(xn, idx1))
    y = rng.random((2, 5, 3)).astype(np.float32)
    d1, i1, d2, i2 = D.nn_distance(T(xn, dev), T(y, dev))
    o = O.nn_distance(xn, y)
    assert np.array_equal(N(d1), o[0]) and np.array_equal(N(i1), o[1]) and np.array_equal(N(d2), o[2]) and np.array_equal(N(i2), o[3])
    dd, ii = I.three_nn(T(xn, dev), T(y[:, :2], dev))                     # fewer than three candidates: +inf / index 0 fill
    od, oi = O.three_nn(xn, y[:, :2])
    assert np.array_equal(N(ii), oi) and np.array_equal(N(dd), od)


@pytest.mark.parametrize("n,m,c,k", [(513, 100, 24, 17), (1024, 1024, 48, 17), (2000, 300, 48, 9), (4096, 64, 24, 32), (1024, 50, 64, 17)])
def test_knn_point_2_chunked_path(ops, dev, n, m, c, k):
    """Feature-space k-NN on clouds of 513 .. 4096 points (the dense blocks of the second 16x pass work on 1024-point patches):
    per-chunk wave kernel + merge == tf.nn.top_k over the full distance matrix (tf_grouping.py:95-114), duplicates included."""
    rng = np.random.default_rng(n + c)
    a = rng.standard_normal((2, n, c)).astype(np.float32)
    a[:, 600 % n] = a[:, 3]
    a[:, n - 1] = a[:, 3]                                   # equal distances in different chunks: lower index first
    q = np.concatenate([a[:, : m // 2], rng.standard_normal((2, m - m // 2, c)).astype(np.float32)], 1)
    d2, i2 = ops["G"].knn_point_2(k, T(a, dev), T(q, dev))
    od, oi2 = O.knn_point_2(k, a, q)
    assert np.array_equal(N(i2), oi2) and np.array_equal(N(d2), od)


@pytest.mark.parametrize("n,m,c,k", [(513, 100, 24, 17), (600, 64, 48, 17), (768, 130, 30, 9), (1000, 1000, 7, 17), (1024, 1024, 48, 17),
                                     (1024, 65, 24, 33), (1023, 200, 25, 64), (1024, 63, 3, 1)])
def test_knn_point_2_single_pass_1024(ops, dev, n, m, c, k):
    """Feature-space k-NN on clouds of 513 .. 1024 points with up to 48 channels: one pass (dot products accumulated over two channel
    halves on the matrix pipe, one selection per query) == tf.nn.top_k over the full matrix (tf_grouping.py:95-114).  Ragged last
    candidate tile / query tile / channel quad, the channel halves (c <= 24 takes one), duplicates, k up to 64."""
    rng = np.random.default_rng(n * 3 + c)
    a = rng.standard_normal((3, n, c)).astype(np.float32)
    a[:, 600 % n] = a[:, 3]
    a[:, n - 1] = a[:, 3]
    q = np.concatenate([a[:, : m // 2], rng.standard_normal((3, m - m // 2, c)).astype(np.float32)], 1)
    d2, i2 = ops["G"].knn_point_2(k, T(a, dev), T(q, dev))
    od, oi2 = O.knn_point_2(k, a, q)
    assert np.array_equal(N(i2), oi2) and np.array_equal(N(d2), od)


def test_knn_point_2_single_pass_degenerate_clouds(ops, dev):
    """More than 128 candidates at exactly the k-th distance (a cloud of few distinct feature rows): the threshold prefilter gives
    up and the full sort answers -- index order within equal distances, as tf.nn.top_k."""
    rng = np.random.default_rng(77)
    base = rng.standard_normal((2, 5, 48)).astype(np.float32)
    a = base[:, rng.integers(0, 5, 1024)]                                    # every row is one of five
    a[0, :300] = 0.0
    q = np.concatenate([a[:, :40], rng.standard_normal((2, 30, 48)).astype(np.float32)], 1)
    for k in (17, 40):
        d2, i2 = ops["G"].knn_point_2(k, T(a, dev), T(q, dev))
        od, oi2 = O.knn_point_2(k, a, q)
        assert np.array_equal(N(i2), oi2) and np.array_equal(N(d2), od)


def test_knn_feat_strided_single_pass_column_slice(ops, dev):
    """The dense blocks hand the k-NN a COLUMN SLICE of the wide feature buffer (row stride 120 floats, 48 channels from
    column 24; no distances wanted): same indices as the packed call, on a 1024-point cloud."""
    from dispu_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(9)
    wide = rng.standard_normal((4, 1024, 120)).astype(np.float32)
    tw = T(wide, dev)
    idx = torch.empty(4, 1024, 17, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(max(1, L.dispu_knn_feat_scratch_bytes(4, 1024, 1024, 48, 17)), dtype=torch.uint8, device=dev)
    _lib.check(L.dispu_knn_feat_strided_ws(4, 1024, 1024, 48, 17, tw.data_ptr() + 24 * 4, 120, tw.data_ptr() + 24 * 4, 120, None,
                                           idx.data_ptr(), ws.data_ptr(), ws.numel(), st), "knn_feat")
    _, oi = O.knn_point_2(17, np.ascontiguousarray(wide[:, :, 24:72]), np.ascontiguousarray(wide[:, :, 24:72]))
    assert np.array_equal(N(idx), oi[..., 1])


# ---------------------------------------------------------------- round 3: optional ops, reference-signature entries ----
def test_select_top_k_vs_reference_golden_and_oracle(ops, dev, golden_dir):
    """tf_grouping.select_top_k (SelectionSort): the reference's own known answer (selection_sort.cpp:65-94, golden generated by
    the compiled reference) and seeded rows with ties / negative zero / k >= n against the oracle, bit for bit."""
    z = g(golden_dir, "ref_selection_sort.npz")
    oi, o = ops["G"].select_top_k(int(z["k"]), T(z["dist"], dev))
    assert np.array_equal(N(oi), z["outi"]) and np.array_equal(N(o), z["out"])
    rng = np.random.default_rng(5)
    for (b, m, n, k) in [(2, 3, 1, 1), (1, 5, 7, 3), (3, 17, 64, 16), (2, 9, 200, 5), (1, 4, 1000, 40), (2, 2, 130, 500)]:
        d = rng.random((b, m, n)).astype(np.float32)
        d[..., ::3] = np.round(d[..., ::3], 1)                 # plenty of exact ties: the FIRST minimum must win
        if n > 4:
            d[0, 0, 1], d[0, 0, 3] = 0.0, -0.0                # -0.0 == 0.0 for the compare: position 1 wins, the values move as they are
        oi, o = ops["G"].select_top_k(k, T(d, dev))
        ri, ro = O.select_top_k(k, d)
        assert np.array_equal(N(oi), ri), (b, m, n, k)
        assert np.array_equal(N(o).view(np.uint32), ro.view(np.uint32)), (b, m, n, k)
    with pytest.raises(ValueError):
        ops["G"].select_top_k(0, T(np.zeros((1, 1, 4), np.float32), dev))


@pytest.mark.parametrize("b,n,m", [(1, 1, 5), (2, 2, 3), (3, 3, 9), (2, 5, 40), (2, 64, 100), (3, 1000, 257), (2, 8192, 300), (2, 8193, 300),
                                   (1, 20000, 1000), (2, 16387, 64)])
def test_prob_sample_index_exact(ops, dev, b, n, m):
    """tf_sampling.prob_sample: cumulative sums bit-identical to the oracle's restatement of cumsumKernel's association (chunks
    of 8192, quads, work-efficient scan, compensated carry), indices exact; plus oracle-independent checks: the sums are within
    fp32 rounding of a float64 cumsum and the indices are numpy.searchsorted on them."""
    from dispu_amd import _lib
    rng = np.random.default_rng(n * 7 + m)
    w = rng.random((b, n)).astype(np.float32)
    r = rng.random((b, m)).astype(np.float32)
    r[:, 0], r[:, -1] = 0.0, 1.0
    ro, rt = O.prob_sample(w, r, return_temp=True)
    out = ops["S"].prob_sample(T(w, dev), T(r, dev))
    assert np.array_equal(N(out), ro)
    L = _lib.lib()
    tw, tr = T(w, dev), T(r, dev)
    temp = torch.empty((b, n), dtype=torch.float32, device=dev)
    o2 = torch.empty((b, m), dtype=torch.int32, device=dev)
    _lib.check(L.dispu_prob_sample(b, n, m, tw.data_ptr(), tr.data_ptr(), temp.data_ptr(), o2.data_ptr(), _lib.stream_ptr(dev)), "prob_sample")
    assert np.array_equal(N(temp), rt) and np.array_equal(N(o2), ro)
    ref = np.cumsum(w.astype(np.float64), axis=1)
    assert np.abs(N(temp) - ref).max() <= 4e-7 * ref.max()
    for i in range(b):
        q = r[i] * N(temp)[i, -1]
        assert np.array_equal(N(o2)[i], np.minimum(np.searchsorted(N(temp)[i], q, "left"), n - 1))


def test_fps_reference_signature_with_reference_sized_temp(dev):
    """dispu_fps keeps the reference launcher's contract: temp is the op's {32, n} allocation (tf_sampling.cpp:115) whatever b
    is.  b = 40 clouds at a region-skipping size: nothing behind 32*n floats is written (canary), indices equal the oracle's and
    dispu_fps_ws's (full-size scratch, too-small scratch, no scratch)."""
    from dispu_amd import _lib
    L = _lib.lib()
    b, n, m = 40, 5000, 70
    x = np.random.default_rng(3).random((b, n, 3)).astype(np.float32)
    tx = T(x, dev)
    want = O.farthest_point_sample(m, x, contract=CONTRACT)
    temp = torch.full((32 * n + 4096,), -7.0, dtype=torch.float32, device=dev)
    out = torch.full((b, m), -1, dtype=torch.int32, device=dev)
    _lib.check(L.dispu_fps(b, n, m, tx.data_ptr(), temp.data_ptr(), out.data_ptr(), CONTRACT, _lib.stream_ptr(dev)), "dispu_fps")
    assert np.array_equal(N(out), want)
    assert bool((temp[32 * n:] == -7.0).all()), "dispu_fps wrote behind the reference's {32, n} temp"
    need = L.dispu_fps_scratch_bytes(b, n, m)
    assert need == b * n * 4
    for nbytes in (need, need - 4, 0):
        sc = torch.full((need // 4 + 1024,), -7.0, dtype=torch.float32, device=dev)
        out = torch.full((b, m), -1, dtype=torch.int32, device=dev)
        _lib.check(L.dispu_fps_ws(b, n, m, tx.data_ptr(), sc.data_ptr() if nbytes else None, nbytes, out.data_ptr(), CONTRACT,
                                  _lib.stream_ptr(dev)), "dispu_fps_ws")
        assert np.array_equal(N(out), want), nbytes
        assert bool((sc[need // 4:] == -7.0).all())
        if nbytes < need:
            assert bool((sc == -7.0).all()), "a scratch smaller than dispu_fps_scratch_bytes must not be touched"
    # n > 24576 needs the running distances: refused without them instead of writing out of bounds
    big = torch.rand((1, 25000, 3), device=dev)
    o = torch.empty((1, 8), dtype=torch.int32, device=dev)
    assert L.dispu_fps_ws(1, 25000, 8, big.data_ptr(), None, 0, o.data_ptr(), CONTRACT, _lib.stream_ptr(dev)) != 0


@pytest.mark.parametrize("b,n,m", [(2, 300, 200), (1, 1024, 1024), (3, 129, 513), (1, 2048, 700)])
def test_match_cost_reference_signature_entries(dev, b, n, m):
    """dispu_match_cost / dispu_match_cost_grad (the launcher's own signature, no scratch) against the *_ws fast paths and the
    oracle: cost 1e-5 relative, gradients 3e-5 absolute (grad2 and grad1-per-lane sums are the same arithmetic -> equal)."""
    from dispu_amd import _lib
    L = _lib.lib()
    x1, x2 = synth_patches(b, n, seed=n), synth_patches(b, m, seed=m + 3)
    mo = O.approx_match(x1, x2)
    t1, t2, tm = T(x1, dev), T(x2, dev), T(mo, dev)
    st = _lib.stream_ptr(dev)
    cost = torch.empty((b,), dtype=torch.float32, device=dev)
    _lib.check(L.dispu_match_cost(b, n, m, t1.data_ptr(), t2.data_ptr(), tm.data_ptr(), cost.data_ptr(), CONTRACT, st), "match_cost")
    sc = torch.empty((max(L.dispu_match_cost_scratch_bytes(b, n, m) // 4, 1),), dtype=torch.float32, device=dev)
    cost_ws = torch.empty((b,), dtype=torch.float32, device=dev)
    _lib.check(L.dispu_match_cost_ws(b, n, m, t1.data_ptr(), t2.data_ptr(), tm.data_ptr(), cost_ws.data_ptr(), sc.data_ptr(), CONTRACT, st),
               "match_cost_ws")
    co = O.match_cost(x1, x2, mo)
    assert np.allclose(N(cost), co, rtol=1e-5) and np.allclose(N(cost_ws), co, rtol=1e-5)
    g1, g2 = torch.empty((b, n, 3), device=dev), torch.empty((b, m, 3), device=dev)
    _lib.check(L.dispu_match_cost_grad(b, n, m, t1.data_ptr(), t2.data_ptr(), tm.data_ptr(), g1.data_ptr(), g2.data_ptr(), CONTRACT, st),
               "match_cost_grad")
    sg = torch.empty((max(L.dispu_match_cost_grad_scratch_bytes(b, n, m) // 4, 1),), dtype=torch.float32, device=dev)
    h1, h2 = torch.empty((b, n, 3), device=dev), torch.empty((b, m, 3), device=dev)
    _lib.check(L.dispu_match_cost_grad_ws(b, n, m, t1.data_ptr(), t2.data_ptr(), tm.data_ptr(), h1.data_ptr(), h2.data_ptr(), sg.data_ptr(),
                                          CONTRACT, st), "match_cost_grad_ws")
    o1, o2 = O.match_cost_grad(x1, x2, mo)
    assert np.allclose(N(g1), o1, atol=3e-5) and np.allclose(N(g2), o2, atol=3e-5)
    assert np.allclose(N(h1), o1, atol=3e-5) and np.array_equal(N(h2), N(g2))


@pytest.mark.parametrize("b,n,m", [(2, 128, 128), (3, 100, 128), (2, 77, 50), (2, 64, 128), (1, 128, 1), (4, 16, 16)])
@pytest.mark.parametrize("arith", [PLAIN, CONTRACT])
def test_approx_match_reference_association_bit_exact_at_one_tile(ops, dev, b, n, m, arith):
    """Ties the kernels to the REFERENCE's own summation order with no reassociation in between: for n, m <= 128 one tile holds
    every partner of every point, so each running sum of the auction is a single sequential chain -- pass 1 starting at 1e-9f
    (tf_approxmatch_g.cu:60), passes 2 / 3 at their first term -- which is oracle chunk = 0, the restatement pinned to the
    reference's approxmatch_cpu golden (tests/test_oracle.py).  Bit-exact in pinned-exp mode; and chunk = 128 degenerates to it."""
    rng = np.random.default_rng(1000 * n + m)
    x1, x2 = rng.random((b, n, 3), dtype=np.float32), rng.random((b, m, 3), dtype=np.float32)
    contract = 1 if arith == CONTRACT else 0
    seq = O.approx_match(x1, x2, contract=contract, pinned_exp=True, chunk=0)
    assert np.array_equal(seq, O.approx_match(x1, x2, contract=contract, pinned_exp=True, chunk=O.AM_CHUNK))
    got = N(ops["A"].approx_match(T(x1, dev), T(x2, dev), arith=arith | PINNED_EXP))
    assert np.array_equal(got, seq)


def test_approx_match_4096_against_the_sequential_order(ops, dev):
    """the 2-D tiled auction sums in chunks of 128 partners; its bit-parity oracle (chunk = 128) is the same
    restatement.  Tie it to the REFERENCE's sequential order (oracle chunk = 0 = tf_approxmatch_g.cu's one chain per thread) at
    the size the tiling targets, (1, 4096, 4096): EMD within 1e-5 relative, row / column sums of the plan within 1e-4 of the
    sequential plan's, plan entries 1e-3 absolute."""
    n = 4096
    x1, x2 = synth_patches(1, n, seed=41), synth_patches(1, n, seed=42)
    seq = O.approx_match(x1, x2, contract=1, pinned_exp=True, chunk=0)
    got = N(ops["A"].approx_match(T(x1, dev), T(x2, dev), arith=CONTRACT | PINNED_EXP))
    assert np.abs(got - seq).max() < 1e-3
    assert np.abs(got.sum(1) - seq.sum(1)).max() < 1e-4 and np.abs(got.sum(2) - seq.sum(2)).max() < 1e-4
    c_seq = O.match_cost(x1, x2, seq)
    cost = N(ops["A"].match_cost(T(x1, dev), T(x2, dev), T(got, dev)))
    assert np.allclose(cost, c_seq, rtol=1e-5)
    prod = N(ops["A"].match_cost(T(x1, dev), T(x2, dev), ops["A"].approx_match(T(x1, dev), T(x2, dev))))     # hardware exp
    assert np.allclose(prod, c_seq, rtol=1e-5)


@pytest.mark.parametrize("b,n,m", [(3, 300, 200), (2, 1100, 1030), (1, 2048, 512), (40, 64, 96), (2, 1, 5), (1, 1024, 1025)])
@pytest.mark.parametrize("arith", [PLAIN, CONTRACT])
def test_approx_match_reference_signature_with_reference_sized_temp(dev, b, n, m, arith):
    """dispu_approx_match keeps the reference launcher's contract: temp is the op's [b, 2 (n + m)] float allocation
    (tf_approxmatch.cpp:164-170).  Nothing behind it is written (canary), every sum is the reference's sequential chain -- bit-exact
    to the oracle's chunk = 0 order in pinned-exp mode at ANY size -- and the hardware-exp EMD is within 1e-5 of the oracle's.
    dispu_approx_match_ws refuses a scratch smaller than dispu_approx_match_scratch_bytes without touching it or `match`."""
    from dispu_amd import _lib
    L = _lib.lib()
    if min(n, m) < 8:
        rng = np.random.default_rng(b * 7919 + n * 31 + m)
        x1, x2 = rng.random((b, n, 3), dtype=np.float32), rng.random((b, m, 3), dtype=np.float32)
    else:
        x1, x2 = synth_patches(b, n, seed=n + 5), synth_patches(b, m, seed=m + 6)
    t1, t2, st = T(x1, dev), T(x2, dev), _lib.stream_ptr(dev)
    nt = b * 2 * (n + m)
    contract = 1 if arith == CONTRACT else 0
    temp = torch.full((nt + 4096,), -7.0, dtype=torch.float32, device=dev)
    match = torch.full((b, m, n), -3.0, dtype=torch.float32, device=dev)            # the entry zeroes it itself (tf_approxmatch_g.cu:16)
    _lib.check(L.dispu_approx_match(b, n, m, t1.data_ptr(), t2.data_ptr(), match.data_ptr(), temp.data_ptr(), arith | PINNED_EXP, st),
               "dispu_approx_match")
    assert bool((temp[nt:] == -7.0).all()), "dispu_approx_match wrote behind the reference's [b, 2(n+m)] temp"
    assert np.array_equal(N(match), O.approx_match(x1, x2, contract=contract, pinned_exp=True, chunk=0))
    _lib.check(L.dispu_approx_match(b, n, m, t1.data_ptr(), t2.data_ptr(), match.data_ptr(), temp.data_ptr(), arith, st), "dispu_approx_match")
    assert bool((temp[nt:] == -7.0).all())
    mo = O.approx_match(x1, x2, contract=contract)
    assert np.allclose(O.match_cost(x1, x2, N(match)), O.match_cost(x1, x2, mo), rtol=1e-5)
    need = L.dispu_approx_match_scratch_bytes(b, n, m)
    assert need > nt * 4
    sc = torch.full((need // 4 + 1024,), -7.0, dtype=torch.float32, device=dev)
    mw = torch.full((b, m, n), -3.0, dtype=torch.float32, device=dev)
    for nbytes in (need - 4, nt * 4, 0):
        assert L.dispu_approx_match_ws(b, n, m, t1.data_ptr(), t2.data_ptr(), mw.data_ptr(), sc.data_ptr(), nbytes, arith, st) != 0
        assert bool((sc == -7.0).all()) and bool((mw == -3.0).all()), "a refused call must not touch scratch or match"
    _lib.check(L.dispu_approx_match_ws(b, n, m, t1.data_ptr(), t2.data_ptr(), mw.data_ptr(), sc.data_ptr(), need, arith | PINNED_EXP, st),
               "dispu_approx_match_ws")
    assert bool((sc[(need + 3) // 4:] == -7.0).all())
    assert np.array_equal(N(mw), O.approx_match(x1, x2, contract=contract, pinned_exp=True, chunk=O.AM_CHUNK))


@pytest.mark.parametrize("b,n,m", [(4, 1024, 1024), (2, 300, 700), (3, 1, 5), (7, 129, 127)])
def test_approx_match_is_run_to_run_identical(ops, dev, b, n, m):
    """The 22 launches of the auction combine their partial sums in a fixed order (no atomics): two calls give the same bits, in
    hardware-exp and pinned-exp mode.  (Rounds 3 - 4 compared them with a one-launch persistent form, removed in round 5.)"""
    rng = np.random.default_rng(b * 1000 + n)
    x1, x2 = T(rng.random((b, n, 3), dtype=np.float32), dev), T(rng.random((b, m, 3), dtype=np.float32), dev)
    a = (ops["A"].approx_match(x1, x2), ops["A"].approx_match(x1, x2, arith=CONTRACT | PINNED_EXP))
    c = (ops["A"].approx_match(x1, x2), ops["A"].approx_match(x1, x2, arith=CONTRACT | PINNED_EXP))
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])

"""CPU tests: the oracle (oracle/dispu_oracle.c) against the committed golden vectors that were produced
by the REFERENCE's own CPU functions, and -- when oracle/_ref is present -- against those functions
live on fresh seeded inputs.  Bit-exact for indices and (contract=0) distances."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import ref as R


def g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_nn_distance_golden(golden_dir):
    z = g(golden_dir, "ref_nndistance.npz")
    d1, i1, d2, i2 = O.nn_distance(z["xyz1"], z["xyz2"], contract=0)
    assert np.array_equal(i1, z["idx1"]) and np.array_equal(i2, z["idx2"])
    assert np.array_equal(d1, z["dist1"]) and np.array_equal(d2, z["dist2"])
    # the contracted flavour only moves distances by an ulp or so
    c1, j1, c2, j2 = O.nn_distance(z["xyz1"], z["xyz2"], contract=1)
    assert np.allclose(c1, d1, rtol=1e-5, atol=1e-7) and np.allclose(c2, d2, rtol=1e-5, atol=1e-7)
    assert (j1 != i1).mean() < 0.01


def test_interpolate_golden(golden_dir):
    z = g(golden_dir, "ref_interpolate.npz")
    d, i = O.three_nn(z["xyz1"], z["xyz2"], contract=0)
    assert np.array_equal(i, z["idx"]) and np.array_equal(d, z["dist"])
    assert np.array_equal(O.three_interpolate(z["points"], z["idx"], z["weight"]), z["out"])
    assert np.array_equal(O.three_interpolate_grad(z["points"], z["idx"], z["weight"], z["grad_out"]), z["grad_points"])


def test_grouping_golden(golden_dir):
    z = g(golden_dir, "ref_grouping.npz")
    idx, cnt = O.query_ball_point(float(z["radius"]), int(z["nsample"]), z["xyz1"], z["xyz2"], contract=0)
    assert np.array_equal(idx, z["idx"])
    assert cnt.min() >= 1 and cnt.max() <= int(z["nsample"])
    assert np.array_equal(O.group_point(z["points"], z["idx"]), z["out"])
    assert np.array_equal(O.group_point_grad(z["points"], z["idx"], z["grad_out"]), z["grad_points"])


def test_knn_golden(golden_dir):
    z = g(golden_dir, "ref_knn.npz")
    k = int(z["k"])
    assert np.array_equal(O.knn_batch(z["support"], z["support"], k).astype(np.int32), z["idx_self"])
    assert np.array_equal(O.knn_batch(z["support"], z["query"], k).astype(np.int32), z["idx_query"])
    assert np.array_equal(z["idx_self"][:, :, 0], np.broadcast_to(np.arange(512), (4, 512)))  # self first


def test_selection_sort_known_answer(golden_dir):
    """The only deterministic known-answer harness in the reference (selection_sort.cpp:65-94)."""
    z = g(golden_dir, "ref_selection_sort.npz")
    outi, out = O.select_top_k(int(z["k"]), z["dist"])
    assert np.array_equal(out, z["out"]) and np.array_equal(outi, z["outi"])


def test_approxmatch_golden(golden_dir):
    z = g(golden_dir, "ref_approxmatch.npz")
    m = O.approx_match(z["xyz1"], z["xyz2"], contract=1)
    assert np.array_equal(m, z["oracle_match"])                       # regression pin of the restatement
    # structural property of the auction (n == m): every row and column of match sums to 1
    assert np.allclose(m.sum(1), 1.0, atol=2e-5) and np.allclose(m.sum(2), 1.0, atol=2e-5)
    cost = O.match_cost(z["xyz1"], z["xyz2"], m, contract=1)
    # reference matchcost_cpu evaluated on the same match (double accumulation): <= 1e-5 relative
    assert np.allclose(cost, z["ref_matchcost_on_oracle_match"], rtol=1e-5)
    # reference CPU approxmatch is a different algorithm variant (11 levels, double): loose bound only
    assert np.allclose(cost, z["cost_cpu_variant"], rtol=2e-3)
    g1, g2 = O.match_cost_grad(z["xyz1"], z["xyz2"], m, contract=1)
    assert np.allclose(g1, z["ref_grad1_on_oracle_match"], atol=2e-5)
    assert np.allclose(g2, z["ref_grad2_on_oracle_match"], atol=2e-5)


def test_gpu_only_pins(golden_dir):
    z = g(golden_dir, "oracle_gpu_only.npz")
    idx = O.farthest_point_sample(96, z["fps_inp"], contract=1)
    assert np.array_equal(idx, z["fps_idx_contract"])
    assert np.array_equal(O.farthest_point_sample(96, z["fps_inp"], contract=0), z["fps_idx_plain"])
    d, i = O.knn_point_2(17, z["feat"], z["feat"])
    assert np.array_equal(i[..., 1], z["knn2_idx"]) and np.array_equal(d, z["knn2_dist"])


def test_fps_structure():
    """Self-checks available without CUDA: first index 0, no repeats while m <= #distinct
    points, selected min-distances non-increasing; tie rule on exact duplicates."""
    rng = np.random.default_rng(3)
    x = rng.random((2, 600, 3)).astype(np.float32)
    idx = O.farthest_point_sample(100, x)
    assert (idx[:, 0] == 0).all()
    for b in range(2):
        assert len(set(idx[b])) == 100
        sel = x[b, idx[b]]
        mind = [np.min(((sel[:j] - sel[j]) ** 2).sum(-1)) for j in range(1, 100)]
        assert all(mind[j] >= mind[j + 1] - 1e-7 for j in range(len(mind) - 1))
    # reference tie rule: lowest (k mod 512), then lowest k.  Points 3 and 515 (= 3 + 512) tie with 1 and 2:
    pts = np.zeros((1, 600, 3), np.float32)
    pts[0, [1, 2, 3, 515], 0] = 5.0       # four identical far points
    idx = O.farthest_point_sample(2, pts)
    assert idx[0, 1] == 1
    pts = np.zeros((1, 600, 3), np.float32)
    pts[0, [515, 600 - 1], 0] = 5.0       # 515 mod 512 = 3 beats 599 mod 512 = 87
    assert O.farthest_point_sample(2, pts)[0, 1] == 515
    pts = np.zeros((1, 600, 3), np.float32)
    pts[0, [100, 513], 0] = 5.0           # 513 mod 512 = 1 beats 100 although 100 < 513
    assert O.farthest_point_sample(2, pts)[0, 1] == 513


def test_query_ball_semantics():
    """Appendix B.1: strict '<', first nsample in index order, padding with the first hit, rows without
    any hit untouched, only radius[0] read."""
    xyz = np.array([[[0, 0, 0], [1, 0, 0], [0.5, 0, 0], [0.25, 0, 0], [3, 0, 0]]], np.float32)
    q = np.array([[[0, 0, 0], [10, 10, 10]]], np.float32)
    init = np.full((1, 2, 4), -7, np.int32)
    idx, cnt = O.query_ball_point(0.5, 4, xyz, q, idx_init=init)
    assert idx[0, 0].tolist() == [0, 3, 0, 0] and cnt[0, 0] == 2       # 0.5 itself is NOT inside (strict)
    assert idx[0, 1].tolist() == [-7, -7, -7, -7] and cnt[0, 1] == 0   # untouched
    idx, cnt = O.query_ball_point(2.0, 2, xyz, q)
    assert idx[0, 0].tolist() == [0, 1] and cnt[0, 0] == 2             # first two in index order, not nearest


def test_knn_point_negated_and_ties():
    xyz1 = np.array([[[0, 0, 0], [1, 0, 0], [1, 0, 0], [2, 0, 0]]], np.float32)
    val, idx = O.knn_point(3, xyz1, np.array([[[1, 0, 0]]], np.float32))
    assert idx[0, 0].tolist() == [1, 2, 0] and val[0, 0].tolist() == [-0.0, -0.0, -1.0]


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_vs_reference_live(seed):
    rng = np.random.default_rng(seed)
    b, n, m = 2, 257 + seed, 130
    x1 = rng.standard_normal((b, n, 3)).astype(np.float32)
    x2 = rng.standard_normal((b, m, 3)).astype(np.float32)
    d, i = R.nnsearch(x1, x2)
    od, oi, _, _ = O.nn_distance(x1, x2, contract=0)
    assert np.array_equal(d, od) and np.array_equal(i, oi)
    d, i = R.threenn(x1, x2)
    od, oi = O.three_nn(x1, x2, contract=0)
    assert np.array_equal(d, od) and np.array_equal(i, oi)
    u1, u2 = rng.random((b, n, 3)).astype(np.float32), rng.random((b, m, 3)).astype(np.float32)
    assert np.array_equal(R.query_ball_point(0.3, 9, u1, u2), O.query_ball_point(0.3, 9, u1, u2, contract=0)[0])
    assert np.array_equal(R.knn_batch(x1, x2, 8, omp=True), O.knn_batch(x1, x2, 8))


def test_edge_shapes():
    """n in {1,2,3}: three_nn with fewer than three known points reports idx 0 / +inf like the reference's
    1e40 sentinel stored to float; k == n k-NN; single-point FPS."""
    x = np.array([[[0, 0, 0], [1, 1, 1]]], np.float32)
    d, i = O.three_nn(x, x[:, :1])
    assert i[0, 1].tolist() == [0, 0, 0] and d[0, 1, 0] == 3.0 and np.isinf(d[0, 1, 1:]).all()
    assert O.knn_batch(x, x, 2)[0].tolist() == [[0, 1], [1, 0]]
    assert O.farthest_point_sample(1, x).tolist() == [[0]]
    assert O.farthest_point_sample(4, x).tolist() == [[0, 1, 0, 0]]   # exhausted cloud: all temp 0 -> index 0


def test_approxmatch_chunked_order_is_a_reassociation():
    """The MI355X kernels add the auction's row / column sums in pieces of 128 partners (oracle chunk = AM_CHUNK) instead of
    one sequential chain (tf_approxmatch_g.cu:37-55: the reference's order, chunk = 0).  Same algorithm, different
    association: the plan still has unit row / column sums and the EMD agrees to 1e-5 (north-star tolerance)."""
    rng = np.random.default_rng(3)
    for (b, n, m) in [(2, 256, 256), (1, 700, 1000)]:
        x1, x2 = rng.random((b, n, 3), dtype=np.float32), rng.random((b, m, 3), dtype=np.float32)
        seq = O.approx_match(x1, x2)
        chk = O.approx_match(x1, x2, chunk=O.AM_CHUNK)
        assert not np.array_equal(seq, chk)                               # it IS a different association
        if n == m:
            assert np.abs(chk.sum(1) - 1).max() < 1e-5 and np.abs(chk.sum(2) - 1).max() < 1e-5
        assert np.allclose(O.match_cost(x1, x2, chk), O.match_cost(x1, x2, seq), rtol=1e-5)
    # a single chunk (n, m <= 128) differs from the sequential chain only in where pass 1's 1e-9 is added
    x1, x2 = rng.random((1, 100, 3), dtype=np.float32), rng.random((1, 90, 3), dtype=np.float32)
    a, s = O.approx_match(x1, x2, chunk=O.AM_CHUNK), O.approx_match(x1, x2)
    assert np.abs(a - s).max() < 1e-3 and np.allclose(O.match_cost(x1, x2, a), O.match_cost(x1, x2, s), rtol=1e-5)


def test_approxmatch_chunked_golden(golden_dir):
    """oracle_approxmatch_chunk128.npz freezes the chunk-of-128 association of the MI355X EMD kernels (pinned exp: bit-reproducible)."""
    z = g(golden_dir, "oracle_approxmatch_chunk128.npz")
    m = O.approx_match(z["xyz1"], z["xyz2"], contract=1, pinned_exp=True, chunk=int(z["chunk"]))
    assert np.array_equal(m, z["match_pinned"])
    assert np.allclose(O.match_cost(z["xyz1"], z["xyz2"], m, contract=1), z["cost"], rtol=1e-6)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 1000, 8192, 8193, 20000])
def test_prob_sample_oracle_self_checks(n):
    """orc_prob_sample restates cumsumKernel + binarysearchKernel (tf_sampling_g.cu:7-104); no CPU twin exists in the reference
    (parity unpinned vs the .cu), so it is pinned structurally: cumulative sums within fp32 rounding of a float64 cumsum,
    non-decreasing for non-negative weights, and the sampled index is the first position whose cumulative weight reaches
    r * total (numpy.searchsorted on the oracle's own sums)."""
    rng = np.random.default_rng(n)
    w = rng.random((3, n), dtype=np.float32)
    r = rng.random((3, 41), dtype=np.float32)
    r[:, 0], r[:, -1] = 0.0, 1.0
    out, temp = O.prob_sample(w, r, return_temp=True)
    ref = np.cumsum(w.astype(np.float64), axis=1)
    assert np.abs(temp - ref).max() <= 4e-7 * ref.max()
    assert (np.diff(temp, axis=1) >= 0).all()
    for i in range(3):
        q = r[i] * temp[i, -1]
        assert np.array_equal(out[i], np.minimum(np.searchsorted(temp[i], q, "left"), n - 1))
    assert (out[:, 0] == 0).all() and (out[:, -1] == n - 1).all() or n == 1

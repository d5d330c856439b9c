"""The data-path oracle (oracle/data.py) against golden vectors produced by the reference's own
Common/point_operation.py (tests/golden/make_golden.py:point_operation_golden, authoring container), plus the
Fetcher / evaluator restatements' own invariants."""
import os

import numpy as np

from oracle import data as D


def test_point_operation_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_point_operation.npz"))
    np.random.seed(int(g["seed"]))
    idx = np.array(D.nonuniform_sampling(1024, 256))
    jit = D.jitter_perturbation_point_cloud(g["x"].copy(), sigma=0.01, clip=0.03)
    rx, rgt = D.rotate_point_cloud_and_gt(jit.copy(), g["gt"].copy())
    sx, sgt, sc = D.random_scale_point_cloud_and_gt(rx.copy(), rgt.copy(), 0.8, 1.2)
    hx, hgt = D.shift_point_cloud_and_gt(sx.copy(), sgt.copy(), 0.3)
    for name, got in (("idx", idx), ("jit", jit), ("rx", rx), ("rgt", rgt), ("sx", sx), ("sgt", sgt), ("scales", sc),
                      ("hx", hx), ("hgt", hgt)):
        assert np.array_equal(got, g[name]), name
    assert len(set(idx.tolist())) == 256 and idx.min() >= 0 and idx.max() < 1024


def test_fetcher_off_by_one_and_shapes():
    from dispu_amd import synth
    gt = synth.patches(10, 1024, seed=3)
    np.random.seed(7)
    f = D.Fetcher(gt, batch_size=4)
    assert f.num_batches == 3 and f.has_next_batch()
    x, g, r = f.next_batch()                     # rows 4..8 of the shuffled set: batch 0 is never served
    assert x.shape == (4, 256, 3) and g.shape == (4, 1024, 3) and r.shape == (4,)
    # scale in [0.8, 1.2] and a pure z rotation: the ground truth's z extent scales, its norm ratio is the scale
    ratio = np.linalg.norm(g.reshape(4, -1), axis=1) / np.linalg.norm(f.gt_data[4:8].reshape(4, -1), axis=1)
    assert np.all(ratio > 0.8 - 1e-9) and np.all(ratio < 1.2 + 1e-9)
    assert np.allclose(g[..., 2], f.gt_data[4:8][..., 2] * ratio[:, None])


def test_evaluate_pair_properties():
    from dispu_amd import synth
    a = synth.patches(1, 300, seed=1)[0].astype(np.float64)
    cd, hd = D.evaluate_pair(a, a)
    assert cd == 0.0 and hd == 0.0
    b = a * 3.0 + 5.0                            # both clouds are normalised first: similarity transforms cancel
    cd2, hd2 = D.evaluate_pair(a, b)
    assert cd2 < 1e-25 and hd2 < 1e-25
    c = synth.patches(1, 400, seed=2)[0].astype(np.float64)
    cd3, hd3 = D.evaluate_pair(a, c)
    assert 0 < cd3 < hd3

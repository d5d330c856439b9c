"""CPU checks of the training oracle (oracle/train_oracle.py): it must agree with the pinned fp32 inference oracle
where the two overlap, its autograd gradients must match finite differences (the reference's own gradient tests
are finite-difference checks with a 1e-4 bound, tf_grouping_op_test.py:23-25, tf_interpolate_op_test.py:19-21),
and the schedule / Adam restatements must reproduce DisPU/model.py:52-54,160-178."""
import math

import numpy as np
import torch

from oracle import generator as G
from oracle import train_oracle as T


def _patch(b=1, seed=2):
    import dispu_amd.synth as S
    return S.patch_with_gt(b, 256, 1024, seed=seed)


def test_eval_forward_matches_fp32_oracle():
    P = G.init_params(1234, bias_scale=0.05, bn_random=True)
    x, _ = _patch()
    idx = T.neighbour_indices(P, x)
    c, f = T.generator_forward(T.to_torch(P, False), x, idx, is_training=False)
    c0, f0 = G.generator_forward(P, x)
    assert np.abs(c.numpy() - c0).max() <= 2e-6 and np.abs(f.numpy() - f0).max() <= 2e-6


def test_gradients_match_finite_differences():
    P = G.init_params(1234, bias_scale=0.05, bn_random=True)
    x, gt = _patch()
    radius = np.ones(1, np.float32)
    idx = T.neighbour_indices(P, x)
    gt_t = torch.as_tensor(gt, dtype=T.DT)

    def loss_of(Pd):
        Pt = T.to_torch(Pd, False)
        c, f = T.generator_forward(Pt, x, idx, True, {})
        return float(T.pu_loss(c, f, gt_t, radius, 0)[0])

    _, _, grads, _, _ = T.loss_and_grads(P, x, gt, radius, 0)
    rng = np.random.default_rng(0)
    for name in ("generator/feature_extraction_coarse/layer2/l1/weights", "generator/upshuffle_0/conv1/weights",
                 "refine/PointShuffle/after_conv/weights", "refine/PointShuffle/weight_net/wconv0/bn/gamma",
                 "refine/fine_coordinate_regressor/fc_layer1/biases"):
        d = rng.standard_normal(P[name].shape)
        d /= np.linalg.norm(d)
        an = float((grads[name] * d).sum())
        errs = []
        for eps in (1e-5, 1e-6):          # the loss is piecewise smooth: a Chamfer arg-min / ReLU flip inside one
            Pp, Pm = dict(P), dict(P)     # +-eps interval spoils that difference quotient, not the other
            Pp[name] = P[name].astype(np.float64) + eps * d
            Pm[name] = P[name].astype(np.float64) - eps * d
            errs.append(abs((loss_of(Pp) - loss_of(Pm)) / (2 * eps) - an))
        assert min(errs) <= 1e-4 * max(abs(an), 1e-3), (name, an, errs)


def test_training_bn_uses_batch_statistics_and_updates_moving_averages():
    P = G.init_params(1234, bn_random=True)
    x, gt = _patch()
    _, _, _, bn, _ = T.loss_and_grads(P, x, gt, np.ones(1, np.float32), 0)
    mm0 = P[T.BN_SCOPE + "moving_mean"].astype(np.float64)
    assert bn["moving_mean"].shape == (16,) and not np.allclose(bn["moving_mean"], mm0)
    # decay 0.95: the new moving mean lies 5 % of the way from the old one to the batch mean
    assert np.all(np.abs(bn["moving_mean"] - 0.95 * mm0) <= 0.05 * 10.0)


def test_schedules():
    assert [T.weight_fine(e) for e in (0, 10, 11, 20, 21, 30, 31)] == [0.01, 0.01, 0.1, 0.1, 0.5, 0.5, 1.0]
    assert T.learning_rate(0) == 1e-3 and T.learning_rate(29) == 1e-3
    assert math.isclose(T.learning_rate(30), 7e-4) and math.isclose(T.learning_rate(61), 1e-3 * 0.49)
    assert T.learning_rate(10 ** 4) == 1e-6


def test_adam_first_step_is_lr_sign():
    P = {"w": np.array([1.0, -2.0, 3.0])}
    out = T.adam_step(P, {"w": np.array([0.5, -4.0, 0.1])}, {}, 1e-3)
    assert np.allclose(out["w"] - P["w"], [-1e-3, 1e-3, -1e-3], rtol=1e-4)

"""SURVEY 8(f3) closed on the device: a TF1 tensor bundle under the REFERENCE's variable names and shapes -> device tensors ->
the generator's output equals the oracle's on the same weights.

Reference sites: `Model.test` restores with tf.train.Saver through Common/model_utils.py:132-139 (DisPU/model.py:350-353),
`Model.train` restores / saves the train graph (model.py:184-190,226: generator variables + Adam slots + beta powers +
`epoch` + `global_step`, file `model-<epoch>` + `checkpoint` state file); variable names come from DisPU/generator.py:45,60
(outer `generator/`, inner `generator` / `refine` scopes) and Common/tf_util.py:87-105,155-176 (4-D conv2d kernels
[1,1,C_in,C_out], 3-D conv1d kernels, `bn/{beta,gamma,moving_mean,moving_variance}`).

The bundle bytes come from tests/tf_bundle_fixture.py -- an assembler written independently of dis-pu_amd/checkpoint.py
after TensorFlow's table builder (separator keys, 16-entry restart runs with shared prefixes, several data blocks, several
shards) -- and, second, from the product's own writer."""
import numpy as np
import pytest
import torch

import tf_bundle_fixture as TFB
from oracle import generator as OG

pytestmark = pytest.mark.gpu


def N(t):
    return t.detach().cpu().numpy()


def _weights(seed):
    # nothing left at its initial value: biases non-zero, the weight net's BN fold non-trivial
    return OG.init_params(seed=seed, bias_scale=0.05, bn_random=True)


@pytest.mark.parametrize("num_shards,block_size", [(1, 262144), (3, 700)])
def test_reference_named_bundle_to_device_matches_oracle(tmp_path, dev, num_shards, block_size):
    from dispu_amd import checkpoint as CK, params as PP, synth
    P = _weights(4321)
    T = TFB.reference_named_variables(P, PP.layer_shapes(), adam=True, epoch=40.0, global_step=4000, adam_t=3999,
                                      rng=np.random.default_rng(1))
    # the names / shapes the reference's Saver would hold (spot checks of each kind)
    assert T["generator/generator/feature_extraction_coarse/layer0/weights"].shape == (1, 1, 3, 24)
    assert T["generator/generator/feature_extraction_coarse/layer2_prep/weights"].shape == (1, 120, 48)
    assert T["generator/refine/PointShuffle/after_conv/weights"].shape == (1, 128, 16, 256)
    assert T["generator/refine/PointShuffle/weight_net/wconv0/bn/moving_variance"].shape == (16,)
    assert T["generator/refine/fine_coordinate_regressor/fc_layer2/weights/Adam_1"].shape == (1, 64, 3)
    facts = TFB.write_tf_style_bundle(str(tmp_path / "model-40"), T, num_shards=num_shards, block_size=block_size)
    TFB.write_checkpoint_state(str(tmp_path), "model-40")
    assert facts["max_shared"] > 0 and (block_size > 4096 or facts["data_blocks"] >= 2)

    epoch, gen = CK.restore_generator(str(tmp_path), device=dev)        # Model.test's restore
    assert epoch == 40
    x = synth.patches(3, 256, seed=77)
    coarse, fine = gen(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    oc, of = OG.generator_forward(P, x)
    assert np.array_equal(N(coarse), oc), "coarse cloud from restored weights differs from the oracle (bit-exact expected)"
    err = float(np.abs(N(fine) - of).max())
    assert err <= 1e-5, "fine cloud from restored weights off by %g" % err
    # every device tensor holds the checkpoint's bytes (kernels flattened [kh*kw*C_in, C_out])
    for k, v in P.items():
        assert np.array_equal(N(gen.P[k]), np.asarray(v, np.float32).reshape(N(gen.P[k]).shape)), k


def test_product_writer_bundle_to_device_matches_oracle(tmp_path, dev):
    from dispu_amd import checkpoint as CK, synth
    P = _weights(99)
    CK.save_generator_params(str(tmp_path / "model"), P, step=7, adam_slots=True)
    epoch, gen = CK.restore_generator(str(tmp_path), device=dev)
    assert epoch == 7
    x = synth.patches(2, 256, seed=5)
    coarse, fine = gen(torch.from_numpy(x).to(dev))
    oc, of = OG.generator_forward(P, x)
    assert np.array_equal(N(coarse), oc) and float(np.abs(N(fine) - of).max()) <= 1e-5


def test_train_state_restore_resumes_the_same_trajectory(tmp_path, dev):
    """Model.train with opts.restore: a Trainer restored from a TRAIN-graph checkpoint (Adam moments, beta powers, epoch,
    global_step) takes the same next steps as the Trainer that wrote it -- parameters, moving statistics and both moments."""
    from dispu_amd import checkpoint as CK, synth
    from dispu_amd.train import Trainer
    P = _weights(5)
    B, n = 4, 256
    rng = np.random.default_rng(3)
    x = torch.from_numpy(synth.patches(B, n, seed=1)).to(dev)
    gt = torch.from_numpy(synth.patches(B, 4 * n, seed=2)).to(dev)
    radius = torch.ones(B, device=dev)
    a = Trainer(params=P, device=dev)
    a.epoch = 12
    for _ in range(3):
        a.train_step(x, gt, radius)
    prefix = CK.save_train_state(str(tmp_path), a)
    assert prefix.endswith("model-12")
    raw = CK.read_bundle(prefix)
    assert abs(float(raw["beta1_power"]) - 0.9 ** 4) < 1e-6 and int(raw["global_step"]) == 3 and float(raw["epoch"]) == 12.0
    assert raw["generator/refine/PointShuffle/after_conv/weights/Adam"].shape == (1, 128, 16, 256)
    assert np.abs(raw["generator/refine/PointShuffle/after_conv/weights/Adam_1"]).max() > 0

    b = Trainer(params=_weights(6), device=dev)          # different weights: everything must come from the checkpoint
    assert CK.restore_train_state(str(tmp_path), b) == 12
    assert b.adam_t == 3 and b.epoch == 12 and b.global_step == 3
    assert torch.equal(a.flat_p, b.flat_p) and torch.equal(a.flat_m, b.flat_m) and torch.equal(a.flat_v, b.flat_v)
    assert torch.equal(a.moving_mean, b.moving_mean) and torch.equal(a.moving_var, b.moving_var)
    for _ in range(2):
        ta = a.train_step(x, gt, radius)
        tb = b.train_step(x, gt, radius)
    torch.cuda.synchronize()
    # the scatter gradients use float atomics (order-free): the two trajectories agree to rounding, not bit for bit
    # (Adam's update is lr * m / sqrt(v): where a gradient entry is ~0 its atomics noise decides the SIGN of a full lr-sized move,
    # so single weights may differ by up to 2 steps x lr; everything else agrees to rounding)
    diff = np.abs(N(a.flat_p) - N(b.flat_p))
    assert diff.max() <= 2.5e-3 and np.quantile(diff, 0.999) <= 2e-5 and diff.mean() <= 1e-6, (diff.max(), np.quantile(diff, 0.999), diff.mean())
    assert np.allclose(N(a.moving_var), N(b.moving_var), rtol=1e-5, atol=1e-7)
    for k in ta:
        assert abs(float(ta[k]) - float(tb[k])) <= 1e-4 * max(1.0, abs(float(ta[k]))), k

    # a test-graph checkpoint (no slots) is refused by the train-side restore with a clear message
    CK.save_generator_params(str(tmp_path / "model"), a.params(), step=13)
    with pytest.raises(KeyError):
        CK.restore_train_state(str(tmp_path), b)


def test_train_state_round_trip_far_into_training(tmp_path, dev):
    """beta1_power = 0.9^(t+1) is denormal from t ~ 830 and 0 from ~ 990; the reference saves every 20 epochs (thousands of steps).
    A bundle written at adam_t = 2500 must restore adam_t = 2500 (explicit counter), and one that only carries TF's two power
    variables (as a real TF checkpoint would) must restore the same bias-corrected learning rate -- not adam_t = 0 or 966."""
    from dispu_amd import checkpoint as CK, synth
    from dispu_amd.train import Trainer
    P = _weights(8)
    B, n = 2, 256
    x = torch.from_numpy(synth.patches(B, n, seed=1)).to(dev)
    gt = torch.from_numpy(synth.patches(B, 4 * n, seed=2)).to(dev)
    radius = torch.ones(B, device=dev)
    a = Trainer(params=P, device=dev)
    a.epoch = 40
    a.train_step(x, gt, radius)
    a.adam_t = 2500                                   # as if 2500 updates had been applied
    prefix = CK.save_train_state(str(tmp_path), a)
    raw = CK.read_bundle(prefix)
    assert int(raw[CK.ADAM_T_KEY]) == 2500 and 0.0 < float(raw["beta1_power"]) < 1.2e-38
    b = Trainer(params=_weights(9), device=dev)
    CK.restore_train_state(str(tmp_path), b)
    assert b.adam_t == 2500
    # the same bundle as TensorFlow would have written it: no explicit counter, beta1_power stuck at the smallest denormal
    del raw[CK.ADAM_T_KEY]
    raw["beta1_power"] = np.array(1e-45, np.float32)
    CK.write_bundle(prefix, raw)
    c = Trainer(params=_weights(10), device=dev)
    CK.restore_train_state(str(tmp_path), c)
    assert abs(c.adam_t - 2500) <= 2, c.adam_t
    for t in (a, c):
        t.train_step(x, gt, radius)
    torch.cuda.synchronize()
    diff = np.abs(N(a.flat_p) - N(c.flat_p))
    # (after ONE real step m / sqrt(v) = 0.1 g / sqrt(0.001 g^2) = 3.2, and at t = 2500 the bias correction no longer damps it: where the
    # float atomics flip the sign of a ~0 gradient entry two replicas move 2 x 3.2 x lr apart; everything else agrees to rounding)
    assert diff.max() <= 7e-3 and np.quantile(diff, 0.999) <= 2e-5, (diff.max(), np.quantile(diff, 0.999))

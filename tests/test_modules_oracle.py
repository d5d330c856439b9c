"""CPU sanity of oracle/modules.py (the restated PointNet++ / EdgeConv / loss compositions)."""
import numpy as np

from oracle import modules as OM


def test_losses_on_identical_clouds_are_zero_and_positive_otherwise():
    rng = np.random.default_rng(0)
    a = rng.random((2, 128, 3)).astype(np.float32)
    b = rng.random((2, 128, 3)).astype(np.float32)
    assert OM.chamfer(a, a) == 0.0 and OM.hausdorff_loss(a, a) == 0.0
    assert OM.chamfer(a, b) > 0 and OM.hausdorff_loss(a, b) >= OM.chamfer(a, b) / 2
    assert OM.earth_mover(a, b) > OM.earth_mover(a, a)
    assert 0.0 <= OM.get_repulsion_loss(a, radius=0.2) <= 0.001


def test_sa_module_shapes_and_min_pool_sign():
    rng = np.random.default_rng(1)
    xyz, pts = rng.random((1, 100, 3)).astype(np.float32), rng.standard_normal((1, 100, 4)).astype(np.float32)
    P = {"s/conv0/weights": rng.standard_normal((7, 8)).astype(np.float32), "s/conv0/biases": np.zeros(8, np.float32)}
    _, mx, idx = OM.pointnet_sa_module(P, "s", xyz, pts, 10, 0.5, 6, [8], None, False, bn=False, pooling="max")
    _, mn, _ = OM.pointnet_sa_module(P, "s", xyz, pts, 10, 0.5, 6, [8], None, False, bn=False, pooling="min")
    assert mx.shape == (1, 10, 8) and idx.shape == (1, 10, 6)
    assert (mn <= 0).all()            # relu output >= 0, "min" pooling of the reference returns max(-x) <= 0

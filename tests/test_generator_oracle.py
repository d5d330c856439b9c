"""CPU tests of the generator oracle (oracle/generator.py): architecture bookkeeping of SURVEY Appendix A."""
import numpy as np

from oracle import generator as OG


def test_parameter_count_and_shapes():
    P = OG.init_params()
    assert OG.num_params(P) + 4 * 16 == 1046998          # SURVEY Appendix A (incl. the 4x16 BatchNorm vectors)
    assert P["refine/PointShuffle/after_conv/weights"].shape == (2048, 256)
    assert P["generator/upshuffle_0/conv1/weights"].shape == (482, 256)
    assert np.allclose(OG.gen_grid(4), [[-.2, -.2], [.2, -.2], [-.2, .2], [.2, .2]])


def test_forward_small():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 64, 3)).astype(np.float32) * 0.3
    P = OG.init_params()
    tap = {}
    c, f = OG.generator_forward(P, x, tap)
    assert c.shape == (1, 256, 3) and f.shape == (1, 256, 3) and np.isfinite(f).all()
    assert np.abs(f - c).max() <= 0.5                     # offset = sigmoid - 0.5
    assert tap["feat480"].shape == (1, 64, 480)
    c2, f2 = OG.generator_forward(P, x)
    assert np.array_equal(c, c2) and np.array_equal(f, f2)


def test_linear_is_an_fma_chain():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((5, 37)).astype(np.float32)
    w = rng.standard_normal((37, 6)).astype(np.float32)
    b = rng.standard_normal(6).astype(np.float32)
    got = OG.linear(x, w, b, relu=True)
    want = np.zeros((5, 6), np.float32)
    for r in range(5):
        for o in range(6):
            acc = np.float32(0)
            for k in range(37):
                acc = np.float32(np.float64(x[r, k]) * np.float64(w[k, o]) + np.float64(acc))   # exact product, one rounding
            want[r, o] = max(np.float32(acc + b[o]), np.float32(0))
    assert np.array_equal(got, want)


def test_product_initialiser_matches_oracle_inventory():
    """dis-pu_amd/params.py (what bench.py / the trainer use) lists the same variables, shapes and Xavier draws."""
    from dispu_amd import params as PP
    a, b = OG.init_params(1234), PP.init_params(1234)
    assert list(a) == list(b)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert PP.num_params(b) == OG.num_params(a) + 32           # + BN gamma / beta

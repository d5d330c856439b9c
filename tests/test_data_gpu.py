"""GPU parity of the data path (dis-pu_amd/dataset.py) and the evaluator (dis-pu_amd/evaluate.py) against
oracle/data.py (itself pinned to the reference's point_operation.py by golden vectors).  The product draws the same
numpy random numbers in the same order, so seeded batches are comparable element by element (fp32 vs the reference's
float64 arithmetic: 1e-6)."""
import numpy as np
import pytest
import torch

from oracle import data as D

pytestmark = pytest.mark.gpu


def test_fetcher_batches_match_oracle(dev):
    from dispu_amd import synth
    from dispu_amd.dataset import Fetcher
    gt = synth.patches(13, 1024, seed=21)
    np.random.seed(99)
    ref = D.Fetcher(gt, batch_size=4)
    ref_batches = [ref.next_batch() for _ in range(2)]
    np.random.seed(99)
    f = Fetcher(gt, gt, batch_size=4, device=dev)
    assert f.num_batches == ref.num_batches and len(f) == 13
    for rx, rg, rr in ref_batches:
        assert f.has_next_batch()
        x, g, r = f.next_batch()
        assert x.shape == (4, 256, 3) and g.shape == (4, 1024, 3)
        assert np.abs(x.cpu().numpy() - rx).max() <= 2e-6
        assert np.abs(g.cpu().numpy() - rg).max() <= 2e-6
        assert np.array_equal(r.cpu().numpy(), rr.astype(np.float32))
    with pytest.raises(IndexError):          # third batch is short (rows 12..13): the reference's indexing fails there too
        f.next_batch()


def test_fetcher_without_augment_is_a_bit_exact_gather(dev):
    from dispu_amd import synth
    from dispu_amd.dataset import Fetcher
    gt = synth.patches(8, 1024, seed=22)
    np.random.seed(5)
    ref = D.Fetcher(gt, batch_size=4, augment=False, shuffle=False)
    rx, rg, _ = ref.next_batch()
    np.random.seed(5)
    f = Fetcher(gt, gt, batch_size=4, augment=False, shuffle=False, device=dev)
    x, g, _ = f.next_batch()
    assert np.array_equal(x.cpu().numpy(), rx.astype(np.float32)) and np.array_equal(g.cpu().numpy(), rg.astype(np.float32))


def test_augment_kernel_general_rotation_and_shift(dev):
    from dispu_amd import _lib
    rng = np.random.default_rng(1)
    b, n = 3, 500
    x = rng.standard_normal((b, n, 3)).astype(np.float32)
    noise = (rng.standard_normal((b, n, 3)) * 0.01).astype(np.float32)
    rot = np.linalg.qr(rng.standard_normal((b, 3, 3)))[0].astype(np.float32)
    scale = rng.uniform(0.5, 2, b).astype(np.float32)
    shift = rng.uniform(-0.3, 0.3, (b, 3)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    tx, tn, tr, ts, th = t(x), t(noise), t(rot.reshape(b, 9).copy()), t(scale), t(shift)
    out = torch.empty_like(tx)
    _lib.check(_lib.lib().dispu_augment(b, n, _lib.ptr(tx), _lib.ptr(tn), _lib.ptr(tr), _lib.ptr(ts), _lib.ptr(th), _lib.ptr(out),
                                        _lib.stream_ptr(dev)), "dispu_augment")
    ref = np.einsum("bni,bij->bnj", (x + noise).astype(np.float64), rot.astype(np.float64)) * scale[:, None, None] + shift[:, None, :]
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n,m", [(8192, 8192), (1000, 3000)])
def test_evaluate_pair(dev, n, m):
    from dispu_amd import synth
    from dispu_amd.evaluate import evaluate_pair
    a = synth.patches(1, n, seed=31)[0]
    b = (synth.patches(1, m, seed=32)[0] * 1.7 + 0.3).astype(np.float32)
    got = evaluate_pair(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
    cd, hd = D.evaluate_pair(a, b)
    assert abs(got["CD"] - cd) <= 1e-5 * cd and abs(got["hausdorff"] - hd) <= 1e-5 * hd


def test_evaluate_dirs_csv(dev, tmp_path):
    from dispu_amd import synth
    from dispu_amd.evaluate import evaluate_dirs
    (tmp_path / "gt").mkdir()
    (tmp_path / "pred").mkdir()
    for i in range(2):
        np.savetxt(tmp_path / "gt" / ("c%d.xyz" % i), synth.patches(1, 512, seed=40 + i)[0], fmt="%.6f")
        np.savetxt(tmp_path / "pred" / ("c%d.xyz" % i), synth.patches(1, 512, seed=50 + i)[0], fmt="%.6f")
    rows = evaluate_dirs(str(tmp_path / "pred"), str(tmp_path / "gt"))
    assert [r["name"] for r in rows] == ["c0.xyz", "c1.xyz"] and all(r["CD"] > 0 for r in rows)
    lines = (tmp_path / "pred" / "evaluation.csv").read_text().strip().splitlines()
    assert lines[0] == "name,CD,hausdorff" and len(lines) == 4 and lines[-1].startswith("avg,")

"""HDF5 reading (dispu_amd.h5, ctypes over libhdf5) against a fixture written by the genuine HDF5 tools
(tests/golden/make_h5_golden.py: h5import 1.10.6; chunked + gzip, big-endian and int datasets)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
FIX = os.path.join(HERE, "golden", "patches_small.h5")


def _lib_or_skip():
    from dispu_amd import h5
    try:
        return h5, h5.lib()
    except RuntimeError as e:
        pytest.skip(str(e))


def _expected():
    from make_h5_golden import arrays
    return arrays()


def test_h5_datasets_match_generator():
    h5, lib = _lib_or_skip()
    assert lib.version >= (1, 10, 0)
    inp, gt, labels = _expected()
    with h5.File(FIX) as f:
        assert sorted(f.keys()) == ["labels", "poisson_1024", "poisson_256"]
        assert f.shape_dtype("poisson_1024") == ((4, 1024, 3), np.dtype("float32"))
        assert f.shape_dtype("labels") == ((4,), np.dtype("int32"))
        a = f["poisson_1024"]                                  # chunked (1, 1024, 3), deflate level 6
        b = f["poisson_256"]                                   # contiguous, big-endian on disk -> native in memory
        c = f["labels"]
        assert "poisson_256" in f and "poisson_2048" not in f
        with pytest.raises(KeyError):
            f["poisson_2048"]
    assert a.dtype == np.float32 and a.flags.c_contiguous and np.array_equal(a, gt)
    assert b.dtype == np.float32 and b.dtype.isnative and np.array_equal(b, inp)
    assert np.array_equal(c, labels)


def test_h5_rejects_non_hdf5(tmp_path):
    h5, _ = _lib_or_skip()
    p = tmp_path / "not.h5"
    p.write_bytes(b"this is not an HDF5 superblock" * 10)
    with pytest.raises(IOError):
        h5.File(str(p))
    with pytest.raises(FileNotFoundError):
        h5.File(str(tmp_path / "missing.h5"))


def test_load_patches_h5_equals_npz(tmp_path):
    _lib_or_skip()
    from dispu_amd import dataset
    inp, gt, _ = _expected()
    np.savez(tmp_path / "p.npz", poisson_256=inp, poisson_1024=gt)
    for random in (True, False):
        a_in, a_gt = dataset.load_patches(FIX, 256, 1024, random=random)
        b_in, b_gt = dataset.load_patches(str(tmp_path / "p.npz"), 256, 1024, random=random)
        assert np.array_equal(a_in, b_in) and np.array_equal(a_gt, b_gt)
        assert a_in.shape == ((4, 1024, 3) if random else (4, 256, 3))

"""TF1 tensor-bundle reader / writer (dis-pu_amd/checkpoint.py).  The repository ships no checkpoint, so the format
restatement is pinned by known-answer bytes of the published formats (CRC-32C check value, LevelDB varints / footer
magic, a hand-assembled BundleEntryProto) and by the writer <-> reader round trip on the full generator."""
import os
import struct

import numpy as np
import pytest

from dispu_amd import checkpoint as CK
from dispu_amd import params as PP


def test_crc32c_check_value_and_mask():
    assert CK.crc32c(b"123456789") == 0xE3069283            # the CRC-32C (Castagnoli) check value
    assert CK.crc32c(b"") == 0
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert CK.unmask_crc(CK.mask_crc(c)) == c
    assert CK.mask_crc(0) == 0xA282EAD8


def test_varints():
    for n, enc in ((0, b"\x00"), (1, b"\x01"), (127, b"\x7f"), (128, b"\x80\x01"), (300, b"\xac\x02"), (2 ** 32, b"\x80\x80\x80\x80\x10")):
        assert CK.put_varint(n) == enc
        assert CK.get_varint(enc + b"\xff", 0) == (n, len(enc))


def test_bundle_entry_known_bytes():
    # dtype DT_FLOAT(1), shape [3, 24], shard 0, offset 96, size 288, crc32c fixed32 0x01020304
    raw = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x03, 0x12, 0x02, 0x08, 0x18, 0x20, 0x60, 0x28, 0xA0, 0x02,
                 0x35, 0x04, 0x03, 0x02, 0x01])
    e = CK._parse_entry(raw)
    assert e["dtype"] == 1 and e["shape"] == [3, 24] and e["offset"] == 96 and e["size"] == 288 and e["crc32c"] == 0x01020304
    assert CK._encode_entry(1, (3, 24), 0, 96, 288, 0x01020304) == raw


def test_round_trip_small(tmp_path):
    rng = np.random.default_rng(0)
    T = {"a/weights": rng.standard_normal((1, 1, 3, 24)).astype(np.float32), "a/biases": np.zeros(24, np.float32),
         "global_step": np.array(7, np.int64), "z": rng.integers(0, 9, (5,)).astype(np.int32), "d": rng.standard_normal((2, 2))}
    prefix = str(tmp_path / "model-12")
    CK.write_bundle(prefix, T)
    data = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", data[-8:])[0] == 0xDB4775248B80FB57 and len(data) > 48
    back = CK.read_bundle(prefix)
    assert list(back) == sorted(T)
    for k in T:
        assert back[k].dtype == T[k].dtype and back[k].shape == T[k].shape and np.array_equal(back[k], T[k])
    # a flipped payload byte is caught by the per-tensor checksum
    blob = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    blob[10] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(blob))
    with pytest.raises(ValueError):
        CK.read_bundle(prefix)


def test_generator_checkpoint_round_trip(tmp_path):
    P = PP.init_params(1234)
    prefix = str(tmp_path / "model-40")
    CK.save_generator_params(prefix, P)
    # on disk: the reference's names and kernel shapes, many small keys -> several table blocks with prefix compression
    _, entries = CK.read_index(prefix + ".index")
    assert entries["generator/generator/feature_extraction_coarse/layer0/weights"]["shape"] == [1, 1, 3, 24]
    assert entries["generator/refine/PointShuffle/after_conv/weights"]["shape"] == [1, 128, 16, 256]
    assert entries["generator/refine/PointShuffle/skip/weights"]["shape"] == [1, 134, 256]
    assert CK.pre_load_checkpoint(str(tmp_path)) == (40, prefix)
    Q = CK.load_generator_params(prefix)
    assert list(Q) == list(P)
    assert all(np.array_equal(Q[k], P[k]) for k in P)
    assert CK.pre_load_checkpoint(str(tmp_path / "nope")) == (0, None)
    # the graph's non-generator globals travel too (DisPU/model.py:42-45), with the reference's dtypes
    raw = CK.read_bundle(prefix)
    assert raw["epoch"].dtype == np.float32 and raw["epoch"].shape == () and float(raw["epoch"]) == 40.0
    assert raw["global_step"].dtype == np.int32 and int(raw["global_step"]) == 0


def test_checkpoint_prefix_rules_and_adam_slots(tmp_path):
    P = PP.init_params(7)
    with pytest.raises(ValueError):
        CK.save_generator_params(str(tmp_path / "model"), P)          # pre_load_checkpoint could not parse this name
    written = CK.save_generator_params(str(tmp_path / "model"), P, step=12, global_step=345, adam_slots=True)
    assert written.endswith("model-12") and CK.pre_load_checkpoint(str(tmp_path)) == (12, written)
    raw = CK.read_bundle(written)
    assert int(raw["global_step"]) == 345 and float(raw["epoch"]) == 12.0
    assert abs(float(raw["beta1_power"]) - 0.9) < 1e-7 and abs(float(raw["beta2_power"]) - 0.999) < 1e-7
    w = "generator/refine/PointShuffle/after_conv/weights"
    assert raw[w + "/Adam"].shape == raw[w].shape == (1, 128, 16, 256) and not raw[w + "/Adam_1"].any()
    Q = CK.load_generator_params(written)                             # slots and globals are skipped on the way back
    assert list(Q) == list(P) and all(np.array_equal(Q[k], P[k]) for k in P)


def test_optimizer_slots_are_ignored_and_missing_variables_reported(tmp_path):
    P = PP.init_params(1)
    shapes = dict(PP.layer_shapes())
    T = {"generator/" + k: (v.reshape(shapes[k[:-8]]) if k.endswith("/weights") else v) for k, v in P.items()}
    T["generator/generator/upshuffle_0/conv1/weights/Adam"] = np.zeros((1, 1, 482, 256), np.float32)
    T["beta1_power"] = np.array(0.9, np.float32)
    prefix = str(tmp_path / "model-3")
    CK.write_bundle(prefix, T)
    Q = CK.load_generator_params(prefix)
    assert list(Q) == list(P)
    del T["generator/refine/PointShuffle/aggregation/biases"]
    CK.write_bundle(prefix, T)
    with pytest.raises(KeyError):
        CK.load_generator_params(prefix)

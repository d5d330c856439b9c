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


# ------------------------------------------------------------------------------------------------------------------
# bytes the product's own writer never emits: tests/tf_bundle_fixture.py restates TensorFlow's C++ table builder
# (separator keys in the index block, size-estimate block cuts, several shards) without importing checkpoint.py
# ------------------------------------------------------------------------------------------------------------------
import tf_bundle_fixture as TFB  # noqa: E402  (tests/ is on sys.path under pytest's rootdir conftest)


def test_fixture_crc_and_separators_known_answers():
    assert TFB.crc32c(b"123456789") == 0xE3069283 and TFB.masked(0) == 0xA282EAD8
    # 32 zero bytes / 32 0xff bytes: the iSCSI test patterns of RFC 3720 B.4
    assert TFB.crc32c(bytes(32)) == 0x8A9136AA and TFB.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert CK.crc32c(bytes(32)) == 0x8A9136AA and CK.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert TFB.shortest_separator(b"abcdefg", b"abzz") == b"abd"
    assert TFB.shortest_separator(b"generator/a/biases", b"generator/a/weights") == b"generator/a/c"
    assert TFB.shortest_separator(b"abc", b"abcd") == b"abc" and TFB.short_successor(b"\xff\xffq") == b"\xff\xffr"


@pytest.mark.parametrize("num_shards,block_size", [(1, 262144), (2, 1024), (3, 512)])
def test_reader_on_independently_assembled_bundle(tmp_path, num_shards, block_size):
    P = PP.init_params(11)
    rng = np.random.default_rng(5)
    for k in P:                                         # nothing at its initial value: biases, BN statistics too
        P[k] = (P[k] + 0.05 * rng.standard_normal(P[k].shape)).astype(np.float32)
    P[PP.BN_SCOPE + "moving_variance"] = np.abs(P[PP.BN_SCOPE + "moving_variance"]) + 0.5
    T = TFB.reference_named_variables(P, PP.layer_shapes(), adam=True, epoch=40.0, global_step=77, adam_t=9, rng=rng)
    prefix = str(tmp_path / "model-40")
    facts = TFB.write_tf_style_bundle(prefix, T, num_shards=num_shards, block_size=block_size)
    TFB.write_checkpoint_state(str(tmp_path), "model-40")
    assert facts["max_shared"] >= 20                                 # prefix-compressed keys with a long shared part
    if block_size < 262144:
        assert facts["data_blocks"] >= 2
        assert all(b > 0 for b in facts["shard_bytes"])              # every shard really holds tensors
    else:
        assert facts["data_blocks"] == 1                             # TF's default block size: the whole index is one block
    header, entries = CK.read_index(prefix + ".index")
    assert header[1] == num_shards and set(entries) == set(T)
    assert {e["shard_id"] for e in entries.values()} == set(range(num_shards))
    assert entries["epoch"]["shape"] == [] and entries["generator/refine/PointShuffle/after_conv/weights/Adam_1"]["shape"] == [1, 128, 16, 256]
    raw = CK.read_bundle(prefix)
    assert list(raw) == sorted(T, key=lambda s: s.encode())
    for k in T:
        assert raw[k].dtype == T[k].dtype and raw[k].shape == T[k].shape and np.array_equal(raw[k], T[k]), k
    assert CK.pre_load_checkpoint(str(tmp_path)) == (40, prefix)
    Q = CK.load_generator_params(prefix)
    assert list(Q) == list(P) and all(np.array_equal(Q[k], P[k]) for k in P)
    # a corrupted byte in the LAST shard is caught
    last = "%s.data-%05d-of-%05d" % (prefix, num_shards - 1, num_shards)
    blob = bytearray(open(last, "rb").read())
    blob[len(blob) // 2] ^= 0x40
    open(last, "wb").write(bytes(blob))
    with pytest.raises(ValueError):
        CK.read_bundle(prefix)


def test_product_writer_read_back_by_the_independent_parser(tmp_path):
    """the other direction: checkpoint.write_bundle's bytes walked by code that shares nothing with it."""
    import struct as S
    P = PP.init_params(3)
    prefix = CK.save_generator_params(str(tmp_path / "model"), P, step=5, adam_slots=True)
    data = open(prefix + ".index", "rb").read()
    assert S.unpack("<Q", data[-8:])[0] == TFB.TABLE_MAGIC

    def uv(buf, p):
        v = s = 0
        while True:
            b = buf[p]
            p += 1
            v |= (b & 127) << s
            s += 7
            if b < 128:
                return v, p

    def block(off, size):
        body = data[off:off + size]
        assert TFB.masked(TFB.crc32c(data[off:off + size + 1])) == S.unpack_from("<I", data, off + size + 1)[0]
        nrest = S.unpack_from("<I", body, size - 4)[0]
        end, p, key, out = size - 4 - 4 * nrest, 0, b"", []
        while p < end:
            sh, p = uv(body, p)
            ns, p = uv(body, p)
            vl, p = uv(body, p)
            key = key[:sh] + body[p:p + ns]
            out.append((key, body[p + ns:p + ns + vl]))
            p += ns + vl
        return out

    p = len(data) - 48
    _, p = uv(data, p)
    _, p = uv(data, p)
    ioff, p = uv(data, p)
    isz, p = uv(data, p)
    keys = []
    for _, h in block(ioff, isz):
        o, q = uv(h, 0)
        z, q = uv(h, q)
        keys += [k for k, _ in block(o, z)]
    assert keys == sorted(keys) and keys[0] == b"" and len(keys) > 2 * len(P)
    assert b"generator/generator/upshuffle_0/conv1/weights/Adam_1" in keys and b"global_step" in keys


# ------------------------------------------------------------------------------------------------------------------
# Adam's step count behind a train-graph bundle (round-4 advisor: beta1_power underflows long before a real checkpoint)
# ------------------------------------------------------------------------------------------------------------------
def _tf_powers(t, beta1=0.9, beta2=0.999):
    """beta1_power / beta2_power exactly as TF's AdamOptimizer leaves them after t applies: initial value beta, then one fp32
    multiply per apply (training/adam.py:_finish)."""
    b1, b2 = np.float32(beta1), np.float32(beta2)
    p1 = np.full((), b1, np.float32)
    p2 = np.full((), b2, np.float32)
    with np.errstate(under="ignore"):
        for _ in range(t):
            p1 = np.float32(p1 * b1)
            p2 = np.float32(p2 * b2)
    return p1, p2


@pytest.mark.parametrize("t", [0, 1, 3, 9, 500, 827])
def test_adam_steps_exact_while_beta1_power_is_normal(t):
    p1, p2 = _tf_powers(t)
    assert CK.adam_steps_from_bundle({"beta1_power": p1, "beta2_power": p2}, 0.9) == t


@pytest.mark.parametrize("t", [900, 966, 1000, 2000, 4000, 20000, 60000])
def test_adam_steps_from_beta2_power_once_beta1_power_underflowed(t):
    p1, p2 = _tf_powers(t)
    assert float(p1) < float(np.finfo(np.float32).tiny)         # denormal or stuck: useless
    got = CK.adam_steps_from_bundle({"beta1_power": p1, "beta2_power": p2}, 0.9)
    # beta2_power carries the accumulated rounding of t fp32 multiplies: t comes back to a few steps, where it no longer matters --
    # what has to agree is the bias-corrected learning-rate factor Adam derives from it
    assert abs(got - t) <= max(2, t // 2000), (got, t)
    lr_t = lambda n: np.sqrt(1.0 - 0.999 ** n) / (1.0 - 0.9 ** n)
    assert abs(lr_t(got) - lr_t(t)) <= 1e-6 * lr_t(t)
    assert got != 0 and (t == 966 or got != 966)                # the two values the old formula produced for every real checkpoint


def test_adam_steps_saturate_when_both_powers_underflowed_and_explicit_counter_wins():
    p1, p2 = _tf_powers(120000)
    got = CK.adam_steps_from_bundle({"beta1_power": p1, "beta2_power": p2}, 0.9)
    assert got == CK.ADAM_T_SATURATED
    assert np.sqrt(1.0 - 0.999 ** got) / (1.0 - 0.9 ** got) == 1.0
    assert CK.adam_steps_from_bundle({"beta1_power": p1, "beta2_power": p2, CK.ADAM_T_KEY: np.array(123456, np.int64)}, 0.9) == 123456
    assert CK.adam_steps_from_bundle({}, 0.9) == 0             # a test-graph bundle: a fresh optimizer


def test_state_file_lists_every_prefix_like_the_reference_saver(tmp_path):
    """DisPU/model.py:184 is tf.train.Saver(max_to_keep=None): the state file lists every checkpoint still on disk.  A finite limit
    (TF's own behaviour) drops the oldest from the list AND from the disk."""
    P = PP.init_params(2)
    eps = (20, 40, 60, 80, 100, 120)
    for ep in eps:
        CK.save_generator_params(str(tmp_path / "model"), P, step=ep)
    txt = (tmp_path / "checkpoint").read_text().splitlines()
    assert txt[0] == 'model_checkpoint_path: "model-120"'
    assert txt[1:] == ['all_model_checkpoint_paths: "model-%d"' % e for e in eps]
    assert all((tmp_path / ("model-%d.index" % e)).exists() for e in eps)
    assert CK.pre_load_checkpoint(str(tmp_path))[0] == 120
    CK._update_state_file(str(tmp_path), "model-120", max_to_keep=2)
    txt = (tmp_path / "checkpoint").read_text().splitlines()
    assert txt[1:] == ['all_model_checkpoint_paths: "model-%d"' % e for e in (100, 120)]
    left = sorted(f for f in os.listdir(tmp_path) if f.startswith("model-"))
    assert left and all(f.startswith(("model-100.", "model-120.")) for f in left)
    assert CK.pre_load_checkpoint(str(tmp_path))[0] == 120

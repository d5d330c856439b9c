"""TF1 checkpoint import / export for the generator weights.

The reference saves and restores with `tf.train.Saver` (DisPU/model.py:184,190,350-353) and finds the latest file
through `tf.train.get_checkpoint_state` (Common/model_utils.py:132-139).  TensorFlow is not available here, so the
on-disk format is read directly.  It is third-party (not under /root/reference): TensorFlow 1.11 "tensor bundle"
(tensorflow/core/util/tensor_bundle/) = `<prefix>.index`, a LevelDB-format sorted string table whose values are
BundleEntryProto messages, plus `<prefix>.data-0000i-of-0000n` holding the raw little-endian tensor bytes.
Restated from the published formats:
  * table file: blocks of prefix-compressed entries (varint32 shared | non_shared | value_len, key delta, value),
    uint32 restart array + count, 1-byte compression tag + masked crc32c trailer per block, 48-byte footer
    (metaindex handle, index handle, magic 0xdb4775248b80fb57);  TF writes the index uncompressed.
  * key "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version};
    key <variable name> -> BundleEntryProto {1: dtype, 2: shape {2: dim {1: size}}, 3: shard_id, 4: offset, 5: size,
    6: crc32c (masked, fixed32)}.
No reference fixture pins this ("parity unpinned": the repository ships no checkpoint); tests cover the writer/reader
round trip, the CRC-32C check value and hand-assembled known-answer bytes.

Variable names: the reference builds the generator under `Generator(name='generator')` with inner scopes
'generator' and 'refine' (DisPU/generator.py:45,60), i.e. 'generator/generator/feature_extraction_coarse/layer0/weights',
'generator/refine/PointShuffle/...'; this package drops the outer 'generator/'.  Conv kernels [kh, kw, C_in, C_out]
are flattened to [kh*kw*C_in, C_out].  Optimizer slots ('.../Adam', '.../Adam_1', 'beta1_power', ...) are ignored.
"""
import os
import re
import struct
from collections import OrderedDict

import numpy as np

MAGIC = 0xDB4775248B80FB57
DT = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8"), 10: np.dtype("bool")}
DT_CODE = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9, np.dtype("bool"): 10}

# ---------------------------------------------------------------------------------------------- CRC-32C ----
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data):
    """Castagnoli CRC-32C (check value crc32c(b'123456789') == 0xE3069283); plain table walk over python ints
    (about a second per 4 MB -- used once per tensor at import / export time)."""
    t = _crc_table().tolist()
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ----------------------------------------------------------------------------------------------- varints ----
def put_varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


# --------------------------------------------------------------------------------------- protobuf subset ----
def _proto_fields(buf):
    """[(field number, wire type, value)] of one message (varint / fixed32 / fixed64 / length-delimited)."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = get_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 2:
            n, pos = get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((fn, wt, v))
    return out


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for fn, _, v in _proto_fields(buf):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:
            for f2, _, dim in _proto_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, s in _proto_fields(dim):
                        if f3 == 1:
                            size = s
                    e["shape"].append(size)
        elif fn == 3:
            e["shard_id"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6:
            e["crc32c"] = v
        elif fn == 7:
            e["sliced"] = True
    return e


def _encode_entry(dtype_code, shape, shard_id, offset, size, crc_masked):
    dims = b"".join(b"\x12" + put_varint(len(d)) + d for d in (b"\x08" + put_varint(s) for s in shape))
    out = b"\x08" + put_varint(dtype_code) + b"\x12" + put_varint(len(dims)) + dims
    if shard_id:
        out += b"\x18" + put_varint(shard_id)
    if offset:
        out += b"\x20" + put_varint(offset)
    out += b"\x28" + put_varint(size) + b"\x35" + struct.pack("<I", crc_masked)
    return out


# ------------------------------------------------------------------------------------------ table reader ----
def _read_block(data, offset, size):
    raw = data[offset:offset + size]
    tag = data[offset + size]
    if tag != 0:
        raise NotImplementedError("compressed table block (type %d): TF writes checkpoint indices uncompressed" % tag)
    stored = struct.unpack_from("<I", data, offset + size + 1)[0]
    if unmask_crc(stored) != crc32c(raw + bytes([tag])):
        raise ValueError("table block checksum mismatch at offset %d" % offset)
    nrestart = struct.unpack_from("<I", raw, len(raw) - 4)[0]
    end = len(raw) - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = get_varint(raw, pos)
        non_shared, pos = get_varint(raw, pos)
        vlen, pos = get_varint(raw, pos)
        key = key[:shared] + raw[pos:pos + non_shared]
        pos += non_shared
        out.append((key, raw[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(path):
    """`<prefix>.index` -> (header fields, OrderedDict name -> entry dict)."""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != MAGIC:
        raise ValueError("%s is not a table file (bad magic)" % path)
    footer = data[-48:]
    pos = 0
    _, pos = get_varint(footer, pos)          # metaindex handle
    _, pos = get_varint(footer, pos)
    ioff, pos = get_varint(footer, pos)
    isize, pos = get_varint(footer, pos)
    entries, header = OrderedDict(), None
    for _, handle in _read_block(data, ioff, isize):
        boff, p = get_varint(handle, 0)
        bsize, p = get_varint(handle, p)
        for key, val in _read_block(data, boff, bsize):
            if key == b"":
                header = {fn: v for fn, _, v in _proto_fields(val)}
            else:
                entries[key.decode()] = _parse_entry(val)
    return header, entries


def read_bundle(prefix, verify=True):
    """every tensor of the checkpoint `<prefix>` -> OrderedDict name -> ndarray."""
    header, entries = read_index(prefix + ".index")
    if header is not None and header.get(2, 0) != 0:
        raise NotImplementedError("big-endian tensor bundle")
    nshards = (header or {}).get(1, 1)
    shards = {}
    out = OrderedDict()
    for name, e in entries.items():
        if e["sliced"]:
            raise NotImplementedError("partitioned variable %s" % name)
        if e["dtype"] not in DT:
            continue                                   # strings etc.: nothing the generator needs
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, nshards), "rb").read()
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if verify and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("tensor %s: data checksum mismatch" % name)
        out[name] = np.frombuffer(raw, DT[e["dtype"]]).reshape(e["shape"]).copy()
    return out


# ------------------------------------------------------------------------------------------ table writer ----
def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (key, val) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(last), len(key)) and last[shared] == key[shared]:
                shared += 1
        out += put_varint(shared) + put_varint(len(key) - shared) + put_varint(len(val)) + key[shared:] + val
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _emit(f, block):
    off = f.tell()
    f.write(block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
    return put_varint(off) + put_varint(len(block))


def write_bundle(prefix, tensors, block_bytes=4096):
    """name -> ndarray mapping -> `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard, little endian)."""
    names = sorted(tensors, key=lambda s: s.encode())
    items = [(b"", b"\x08\x01\x1a\x02\x08\x01")]                     # header: num_shards 1, version {producer 1}
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as fd:
        for n in names:
            a = np.asarray(tensors[n])
            if a.ndim and not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)
            a = a.astype(a.dtype.newbyteorder("<"), copy=False)
            raw = a.tobytes()
            fd.write(raw)
            items.append((n.encode(), _encode_entry(DT_CODE[np.dtype(a.dtype.name)], a.shape, 0, offset, len(raw),
                                                    mask_crc(crc32c(raw)))))
            offset += len(raw)
    with open(prefix + ".index", "wb") as f:
        index_items, cur, size = [], [], 0
        for kv in items:
            cur.append(kv)
            size += len(kv[0]) + len(kv[1]) + 6
            if size >= block_bytes:
                index_items.append((cur[-1][0], _emit(f, _build_block(cur))))
                cur, size = [], 0
        if cur:
            index_items.append((cur[-1][0], _emit(f, _build_block(cur))))
        meta = _emit(f, _build_block([]))
        index = _emit(f, _build_block(index_items, restart_interval=1))
        footer = meta + index
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))


# ------------------------------------------------------------------------------------- reference-facing ----
def pre_load_checkpoint(checkpoint_dir):
    """Common/model_utils.py:132-139 -> (epoch_step, checkpoint prefix) from the `checkpoint` state file."""
    state = os.path.join(checkpoint_dir, "checkpoint")
    if not os.path.isfile(state):
        return 0, None
    m = re.search(r'^model_checkpoint_path:\s*"(.*)"', open(state).read(), re.M)
    if not m:
        return 0, None
    path = m.group(1)
    if not os.path.isabs(path):
        path = os.path.join(checkpoint_dir, path)
    return int(os.path.basename(path).split("-")[1]), path


def _is_slot(name):
    leaf = name.rsplit("/", 1)[-1]
    return leaf.startswith("Adam") or leaf in ("beta1_power", "beta2_power", "global_step", "epoch")


def load_generator_params(prefix, scope="generator"):
    """checkpoint -> the name -> float32 array mapping Generator.load_params / Trainer.load_params take."""
    from .params import layer_shapes, BN_SCOPE
    raw = read_bundle(prefix)
    P = OrderedDict()
    for name, a in raw.items():
        if not name.startswith(scope + "/") or _is_slot(name):
            continue
        local = name[len(scope) + 1:]
        if local.endswith("/weights"):
            a = a.reshape(-1, a.shape[-1])
        P[local] = np.ascontiguousarray(a, np.float32)
    want = [s + "/" + leaf for s, _ in layer_shapes() for leaf in ("weights", "biases")]
    want += [BN_SCOPE + leaf for leaf in ("gamma", "beta", "moving_mean", "moving_variance")]
    missing = [w for w in want if w not in P]
    if missing:
        raise KeyError("checkpoint %s lacks %d generator variables, e.g. %s" % (prefix, len(missing), missing[:3]))
    for s, shp in layer_shapes():
        if P[s + "/weights"].shape != (int(np.prod(shp[:-1])), shp[-1]):
            raise ValueError("%s/weights has shape %s, expected kernel %s" % (s, P[s + "/weights"].shape, shp))
    return OrderedDict((k, P[k]) for k in want)


def save_generator_params(prefix, params, scope="generator", step=None, epoch=None, global_step=0, adam_slots=False):
    """the inverse of load_generator_params: write the generator's variables under the reference's names and 4-D / 3-D kernel
    shapes, plus the graph's two non-generator globals `epoch` (float32 scalar) and `global_step` (int32 scalar)
    (DisPU/model.py:42-45) and the `checkpoint` state file.

    File name: the reference saves as `<log_dir>/model-<epoch>` (model.py:226) and parses the number after the dash
    (Common/model_utils.py:138), so `step` is appended to `prefix` when given and a prefix without a `-<number>` tail is
    refused.  Returns the prefix written.

    What can restore it: this module's reader, and a tf.train.Saver over the reference's TEST graph (model.py:350-353: the
    generator variables and the two globals are all it holds).  The TRAIN graph's Saver also wants the Adam slots
    (`<var>/Adam`, `<var>/Adam_1`, `beta1_power`, `beta2_power`); `adam_slots=True` writes them as a fresh optimizer state
    (zeros, beta powers 0.9 / 0.999).  No checkpoint written by real TensorFlow is available here: the format is pinned by
    known-answer bytes and round trips only (tests/test_checkpoint.py), "parity unpinned" at the TF boundary."""
    from .params import layer_shapes
    if step is not None:
        prefix = "%s-%d" % (prefix, int(step))
    base = os.path.basename(prefix)
    m = re.search(r"-(\d+)$", base)
    if not m:
        raise ValueError("checkpoint prefix %r needs a -<step> tail (pass step=...): pre_load_checkpoint parses it" % base)
    shapes = dict(layer_shapes())
    out = {}
    for k, v in params.items():
        a = np.asarray(v, np.float32)
        if k.endswith("/weights"):
            a = a.reshape(shapes[k[:-len("/weights")]])
        out[scope + "/" + k] = a
        if adam_slots and k.endswith(("/weights", "/biases", "/gamma", "/beta")):
            out[scope + "/" + k + "/Adam"] = np.zeros_like(a)
            out[scope + "/" + k + "/Adam_1"] = np.zeros_like(a)
    if adam_slots:
        out["beta1_power"] = np.array(0.9, np.float32)
        out["beta2_power"] = np.array(0.999, np.float32)
    out["epoch"] = np.array(float(int(m.group(1)) if epoch is None else epoch), np.float32)
    out["global_step"] = np.array(int(global_step), np.int32)
    write_bundle(prefix, out)
    _update_state_file(os.path.dirname(os.path.abspath(prefix)), base)
    return prefix


# ------------------------------------------------------------------------ the reference's two restore sites ----
def restore_generator(log_dir, device=None, opts=None):
    """`Model.test` (DisPU/model.py:350-353: Saver().restore(sess, pre_load_checkpoint(log_dir))) -> (restore_epoch, Generator):
    the latest checkpoint named by `<log_dir>/checkpoint`, loaded into device tensors."""
    from .generator import Generator
    epoch, prefix = pre_load_checkpoint(log_dir)
    if prefix is None:
        raise FileNotFoundError("no checkpoint state file under %s" % log_dir)
    return epoch, Generator(opts=opts, params=load_generator_params(prefix), device=device)


def restore_train_state(log_dir, trainer):
    """`Model.train` with opts.restore (DisPU/model.py:190-193): parameters, BN moving statistics, BOTH Adam moments, the
    bias-correction powers, `epoch` and `global_step` of the latest checkpoint -> `trainer` (device buffers).  Returns the
    restore epoch the training loop resumes from (Common/model_utils.py:138)."""
    import math
    import torch
    epoch, prefix = pre_load_checkpoint(log_dir)
    if prefix is None:
        raise FileNotFoundError("no checkpoint state file under %s" % log_dir)
    raw = read_bundle(prefix)
    trainer.load_params(load_generator_params(prefix))
    missing = []
    for k in trainer.names:
        for slot, flat in (("Adam", trainer.flat_m), ("Adam_1", trainer.flat_v)):
            name = "generator/%s/%s" % (k, slot)
            if name not in raw:
                missing.append(name)
                continue
            view = trainer.P[k]
            off = (view.data_ptr() - trainer.flat_p.data_ptr()) // 4
            flat[off:off + view.numel()].copy_(torch.from_numpy(np.ascontiguousarray(raw[name], np.float32).reshape(-1)))
    if missing:
        raise KeyError("checkpoint %s lacks %d Adam slots (a test-graph checkpoint?), e.g. %s" % (prefix, len(missing), missing[:2]))
    trainer.adam_t = adam_steps_from_bundle(raw, float(trainer.opts.beta))
    trainer.epoch = int(round(float(raw["epoch"]))) if "epoch" in raw else epoch
    # `global_step`: the reference calls minimize() without it (DisPU/model.py:178), so ITS bundles always hold 0; bundles written
    # by save_train_state hold this repo's own step counter (Trainer.global_step, one per train_step).  Nothing on the compute path
    # reads it (the learning rate follows `epoch`, Adam's bias correction follows adam_t).
    trainer.global_step = int(raw["global_step"]) if "global_step" in raw else 0
    return epoch


ADAM_T_KEY = "dispu/adam_t"       # int64 scalar only this repo's writer emits (a tf.train.Saver ignores names its graph lacks)
ADAM_T_SATURATED = 1 << 30        # both bias corrections are exactly 1.0f from ~1.6e4 steps on: any larger t behaves the same


def adam_steps_from_bundle(raw, beta1, beta2=0.999):
    """Number of Adam updates t behind a train-graph bundle.  TF keeps beta^(t+1) (initial value beta, one fp32 multiply per
    apply) in `beta1_power` / `beta2_power`.  0.9^(t+1) leaves the normal fp32 range at t ~ 828, is denormal until ~ 980 and then
    sticks at a denormal or 0, so beyond a few hundred steps it cannot give t -- the reference saves every 20 epochs (thousands of
    steps).  Order of preference: (1) this repo's explicit counter; (2) beta1_power while it is a NORMAL fp32 (exact: the spacing of
    log(p) between consecutive t is 0.105 against an accumulated rounding error < 1e-4); (3) beta2_power while normal (0.999^(t+1)
    stays normal until t ~ 87 k; its accumulated rounding error blurs t by a few steps only where both corrections already equal 1
    to ~1e-7, i.e. where the value no longer matters); (4) saturated: both corrections are 1."""
    import math
    if ADAM_T_KEY in raw:
        return max(0, int(np.asarray(raw[ADAM_T_KEY]).reshape(-1)[0]))
    tiny = float(np.finfo(np.float32).tiny)
    p1 = float(np.asarray(raw["beta1_power"], np.float32)) if "beta1_power" in raw else None
    p2 = float(np.asarray(raw["beta2_power"], np.float32)) if "beta2_power" in raw else None
    if p1 is not None and p1 >= 1.0 and (p2 is None or p2 >= 1.0):
        return 0
    if p1 is not None and tiny <= p1 < 1.0:
        return max(0, int(round(math.log(p1) / math.log(beta1))) - 1)
    if p2 is not None and tiny <= p2 < 1.0:
        return max(0, int(round(math.log(p2) / math.log(beta2))) - 1)
    if p1 is None and p2 is None:
        return 0
    return ADAM_T_SATURATED


def save_train_state(log_dir, trainer, epoch=None):
    """`self.saver.save(sess, os.path.join(log_dir, 'model'), epoch)` (DisPU/model.py:226) for a Trainer: everything a
    Saver over the reference's TRAIN graph holds, i.e. save_generator_params + the real Adam moments and powers."""
    from .params import layer_shapes
    epoch = trainer.epoch if epoch is None else int(epoch)
    P = trainer.params()
    shapes = dict(layer_shapes())
    out = {}
    for k, v in P.items():
        a = np.asarray(v, np.float32)
        if k.endswith("/weights"):
            a = a.reshape(shapes[k[:-len("/weights")]])
        out["generator/" + k] = a
    for k in trainer.names:
        view = trainer.P[k]
        off = (view.data_ptr() - trainer.flat_p.data_ptr()) // 4
        shp = out["generator/" + k].shape
        out["generator/%s/Adam" % k] = trainer.flat_m[off:off + view.numel()].cpu().numpy().reshape(shp).copy()
        out["generator/%s/Adam_1" % k] = trainer.flat_v[off:off + view.numel()].cpu().numpy().reshape(shp).copy()
    # what TF's variables would hold after adam_t applies (beta^(t+1), flushed to the smallest denormal where TF's repeated
    # fp32 multiply sticks), plus the exact counter under a name of this repo's own
    t1 = min(trainer.adam_t, 1 << 20) + 1
    out["beta1_power"] = np.maximum(np.array(float(trainer.opts.beta) ** t1, np.float32), np.float32(1e-45))
    out["beta2_power"] = np.maximum(np.array(0.999 ** t1, np.float32), np.float32(1e-45))
    out[ADAM_T_KEY] = np.array(int(trainer.adam_t), np.int64)
    out["epoch"] = np.array(float(epoch), np.float32)
    out["global_step"] = np.array(int(trainer.global_step), np.int32)
    os.makedirs(log_dir, exist_ok=True)
    prefix = os.path.join(log_dir, "model-%d" % epoch)
    write_bundle(prefix, out)
    _update_state_file(log_dir, os.path.basename(prefix))
    return prefix


def _update_state_file(log_dir, base, max_to_keep=None):
    """The `checkpoint` state file as the reference's tf.train.Saver(max_to_keep=None) (DisPU/model.py:184) maintains it: the newest
    prefix first as model_checkpoint_path, then EVERY prefix saved so far, oldest first, as all_model_checkpoint_paths -- nothing on
    disk drops out of the list.  A finite `max_to_keep` behaves like TF's: prefixes that fall off the list have their bundle files
    (`<prefix>.index`, `<prefix>.data-*`) deleted too."""
    path = os.path.join(log_dir, "checkpoint")
    prev = []
    if os.path.exists(path):
        with open(path) as f:
            prev = re.findall(r'all_model_checkpoint_paths:\s*"([^"]+)"', f.read())
    keep = [p for p in prev if p != base] + [base]
    if max_to_keep:
        for gone in keep[:-max_to_keep]:
            for f in os.listdir(log_dir):
                if f == gone + ".index" or f.startswith(gone + ".data-"):
                    os.remove(os.path.join(log_dir, f))
        keep = keep[-max_to_keep:]
    with open(path, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % base)
        for p in keep:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)

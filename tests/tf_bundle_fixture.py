"""An independently written assembler of TensorFlow-1.x tensor bundles, for tests only.

dis-pu_amd/checkpoint.py has its own writer; a reader that is only ever fed by that writer proves little.  This file
builds `<prefix>.index` + `<prefix>.data-*` the way TensorFlow's own C++ writer lays them out (restated from the
published formats: tensorflow/core/lib/io/table_builder.cc + block_builder.cc + format.cc -- the LevelDB table format --
and tensorflow/core/util/tensor_bundle/tensor_bundle.cc), deliberately covering what checkpoint.write_bundle never emits:

  * index-block keys are SHORTEST SEPARATORS between a data block's last key and the next block's first key, and a
    short successor after the last block (table_builder.cc: FindShortestSeparator / FindShortSuccessor), not real keys;
  * data blocks are cut by the builder's size estimate (TF's default block_size is 256 KiB -> one block for this model;
    `block_size` small -> many blocks), restart points every 16 entries, prefix-compressed keys in between;
  * several shards (`data-0000i-of-0000n`), entries carrying shard_id and per-shard offsets, as MergeBundles leaves them;
  * proto3 field omission (zero shard_id / offset are absent), scalar shapes as an empty TensorShapeProto.

Nothing here imports dis-pu_amd/checkpoint.py: different code, same format.  It shares no helper with the product, so a
misreading of the format would have to be made twice, in two different ways, to go unnoticed.
"""
import io
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_TF_DTYPE = {"float32": 1, "float64": 2, "int32": 3, "int64": 9, "bool": 10}


# --------------------------------------------------------------------------------------------------- CRC-32C ----
def _crc32c_nibble_table():
    tab = []
    for n in range(16):
        c = n
        for _ in range(4):
            c = (c >> 1) ^ (0x82F63B78 if (c & 1) else 0)
        tab.append(c)
    return tab


_NIB = _crc32c_nibble_table()


def crc32c(buf):
    """reflected Castagnoli CRC, four bits at a time (a different walk from the product's 256-entry table)."""
    c = 0xFFFFFFFF
    nib = _NIB
    for b in bytes(buf):
        c ^= b
        c = (c >> 4) ^ nib[c & 15]
        c = (c >> 4) ^ nib[c & 15]
    return c ^ 0xFFFFFFFF


def masked(c):
    rot = ((c >> 15) | ((c << 17) & 0xFFFFFFFF)) & 0xFFFFFFFF
    return (rot + 0xA282EAD8) & 0xFFFFFFFF


# --------------------------------------------------------------------------------------------------- encoding ----
def varint(n):
    s = io.BytesIO()
    while n >= 0x80:
        s.write(bytes(((n & 0x7F) | 0x80,)))
        n >>= 7
    s.write(bytes((n,)))
    return s.getvalue()


def _tag(field, wire):
    return varint((field << 3) | wire)


def _len_delimited(field, payload):
    return _tag(field, 2) + varint(len(payload)) + payload


def tensor_shape_proto(shape):
    return b"".join(_len_delimited(2, _tag(1, 0) + varint(int(d))) for d in shape)


def bundle_entry_proto(dtype, shape, shard_id, offset, size, crc):
    msg = _tag(1, 0) + varint(dtype) + _len_delimited(2, tensor_shape_proto(shape))
    if shard_id != 0:
        msg += _tag(3, 0) + varint(shard_id)
    if offset != 0:
        msg += _tag(4, 0) + varint(offset)
    if size != 0:
        msg += _tag(5, 0) + varint(size)
    if crc != 0:
        msg += _tag(6, 5) + struct.pack("<I", crc)
    return msg


def bundle_header_proto(num_shards):
    # BundleHeaderProto {num_shards = 1; endianness = 2 (LITTLE = 0: omitted); version = 3 {producer = 1}}
    return _tag(1, 0) + varint(num_shards) + _len_delimited(3, _tag(1, 0) + varint(1))


# ----------------------------------------------------------------------------------------------- table builder ----
def shortest_separator(start, limit):
    """leveldb BytewiseComparator::FindShortestSeparator."""
    n = min(len(start), len(limit))
    i = 0
    while i < n and start[i] == limit[i]:
        i += 1
    if i < n and start[i] < 0xFF and start[i] + 1 < limit[i]:
        return start[:i] + bytes((start[i] + 1,))
    return start


def short_successor(key):
    """leveldb BytewiseComparator::FindShortSuccessor."""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes((b + 1,))
    return key


class _Block(object):
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.reset()

    def reset(self):
        self.buf = io.BytesIO()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b""
        self.n = 0

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            lim = min(len(self.last_key), len(key))
            while shared < lim and self.last_key[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(self.buf.tell())
            self.counter = 0
        self.buf.write(varint(shared))
        self.buf.write(varint(len(key) - shared))
        self.buf.write(varint(len(value)))
        self.buf.write(key[shared:])
        self.buf.write(value)
        self.last_key = key
        self.counter += 1
        self.n += 1
        return shared

    def size_estimate(self):
        return self.buf.tell() + 4 * len(self.restarts) + 4

    def finish(self):
        tail = b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))
        return self.buf.getvalue() + tail


class TableBuilder(object):
    """table_builder.cc restated: Add() keys in order, Flush() by size estimate, pending index entry with a separator key."""

    def __init__(self, fileobj, block_size=262144, restart_interval=16):
        self.f = fileobj
        self.block_size = block_size
        self.data = _Block(restart_interval)
        self.index = _Block(1)
        self.pending = None                       # (last key of the flushed block, its handle)
        self.last_key = b""
        self.offset = 0
        self.num_data_blocks = 0
        self.max_shared = 0

    def _write_raw_block(self, contents):
        trailer = b"\x00"                         # kNoCompression
        crc = masked(crc32c(contents + trailer))
        handle = varint(self.offset) + varint(len(contents))
        blob = contents + trailer + struct.pack("<I", crc)
        self.f.write(blob)
        self.offset += len(blob)
        return handle

    def add(self, key, value):
        assert self.data.n == 0 or key > self.last_key, "keys must be added in increasing order"
        if self.pending is not None:
            sep = shortest_separator(self.pending[0], key)
            self.index.add(sep, self.pending[1])
            self.pending = None
        # bookkeeping for the tests: how much prefix compression this table really contains
        self.max_shared = max(self.max_shared, self.data.add(key, value))
        self.last_key = key
        if self.data.size_estimate() >= self.block_size:
            self.flush()

    def flush(self):
        if self.data.n == 0:
            return
        handle = self._write_raw_block(self.data.finish())
        self.pending = (self.last_key, handle)
        self.data.reset()
        self.num_data_blocks += 1

    def finish(self):
        self.flush()
        meta_handle = self._write_raw_block(_Block(1).finish())
        if self.pending is not None:
            self.index.add(short_successor(self.pending[0]), self.pending[1])
            self.pending = None
        index_handle = self._write_raw_block(self.index.finish())
        footer = meta_handle + index_handle
        footer += b"\x00" * (40 - len(footer))
        self.f.write(footer + struct.pack("<Q", TABLE_MAGIC))


# ------------------------------------------------------------------------------------------------ the bundle ----
def write_tf_style_bundle(prefix, tensors, num_shards=1, block_size=262144):
    """tensors: name -> ndarray.  Variables are dealt to the shards in contiguous key ranges (as a per-device save merged by
    MergeBundles would leave them).  Returns facts about the bytes written that the tests assert on."""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    per = (len(names) + num_shards - 1) // num_shards
    shard_of = {n: min(i // per, num_shards - 1) for i, n in enumerate(names)}
    files = [open("%s.data-%05d-of-%05d" % (prefix, s, num_shards), "wb") for s in range(num_shards)]
    pos = [0] * num_shards
    with open(prefix + ".index", "wb") as fi:
        tb = TableBuilder(fi, block_size=block_size)
        tb.add(b"", bundle_header_proto(num_shards))
        for n in names:
            a = np.asarray(tensors[n])
            raw = np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            s = shard_of[n]
            files[s].write(raw)
            tb.add(n.encode("utf-8"), bundle_entry_proto(_TF_DTYPE[a.dtype.name], a.shape, s, pos[s], len(raw), masked(crc32c(raw))))
            pos[s] += len(raw)
        tb.finish()
    for f in files:
        f.close()
    return dict(data_blocks=tb.num_data_blocks, max_shared=tb.max_shared, shards=num_shards, shard_bytes=pos,
                shard_of=shard_of)


def write_checkpoint_state(directory, base):
    """the text-format CheckpointState proto tf.train.Saver.save leaves next to the bundle (`checkpoint`)."""
    with open(os.path.join(directory, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % base)
        f.write('all_model_checkpoint_paths: "%s"\n' % base)


def reference_named_variables(P, layer_shapes, adam=True, epoch=40.0, global_step=1234, beta1=0.9, beta2=0.999, adam_t=0, rng=None):
    """the variable set a tf.train.Saver over the reference's TRAIN graph holds (DisPU/model.py:42-45,158-190,
    DisPU/generator.py:45,60, Common/tf_util.py:87-105,155-176): `generator/<inner scope>/.../weights` as 4-D conv2d / 3-D conv1d
    kernels, biases, the weight net's contrib batch_norm quartet, Adam's two slots per trainable variable, the two beta powers,
    `epoch` (float32 scalar) and `global_step` (int32 scalar).  P holds flattened [C_in_total, C_out] kernels."""
    shapes = dict(layer_shapes)
    T = {}
    for k, v in P.items():
        a = np.asarray(v, np.float32)
        if k.endswith("/weights"):
            a = a.reshape(shapes[k[:-len("/weights")]])
        T["generator/" + k] = a
        if adam and not k.endswith(("moving_mean", "moving_variance")):
            if rng is None:
                T["generator/" + k + "/Adam"] = np.zeros_like(a)
                T["generator/" + k + "/Adam_1"] = np.zeros_like(a)
            else:
                T["generator/" + k + "/Adam"] = (1e-3 * rng.standard_normal(a.shape)).astype(np.float32)
                T["generator/" + k + "/Adam_1"] = (1e-6 * rng.random(a.shape)).astype(np.float32)
    if adam:
        T["beta1_power"] = np.array(beta1 ** (adam_t + 1), np.float32)      # TF initialises the powers to beta, then multiplies per step
        T["beta2_power"] = np.array(beta2 ** (adam_t + 1), np.float32)
    T["epoch"] = np.array(epoch, np.float32)
    T["global_step"] = np.array(global_step, np.int32)
    return T

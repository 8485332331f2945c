"""TensorFlow checkpoint (tensor-bundle V2) import without TensorFlow.

The reference restores its networks with ``tf.train.Saver().restore(sess, model_path)``
(mvsnet/predictlib.py:69-76) from ``<model_dir>/<regularization>/<network_mode>/model.ckpt-<step>``
(mvsnet/utils.py:75-95).  A V2 checkpoint is two kinds of files:

  ``<prefix>.index``                an immutable sorted string table (the LevelDB table format):
                                    key = variable name, value = serialized ``BundleEntryProto``
                                    (dtype, shape, shard, offset, size, masked crc32c); key "" holds
                                    the ``BundleHeaderProto``
  ``<prefix>.data-0000N-of-0000M``  the raw little-endian tensor bytes

This module reads that format with the standard library + numpy (`read_checkpoint`), writes it
(`write_checkpoint`, used for fixtures and for exporting weights), and maps the reference's variable
names onto the parameter dictionaries of this package (`load_mvsnet_params`).

PARITY UNPINNED: no checkpoint and no TensorFlow exist in the build environment, so the reader has
only been exercised against files produced by `write_checkpoint` from the same format description
(tests/test_tf_checkpoint.py) -- it has not been run on a file written by TensorFlow itself.
Snappy-compressed index blocks and partitioned (sliced) variables raise NotImplementedError.
"""
from __future__ import annotations

import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_MASK_DELTA = 0xA282EAD8

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DTYPE_IDS = {v: k for k, v in _DTYPES.items()}


# ---- crc32c (Castagnoli), masked as in tensorflow/core/lib/hash/crc32c.h -----------------------------
def _make_crc_table():
    poly = 0x82F63B78
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t[i] = c
    return t


_CRC_TABLE = _make_crc_table()


def _crc_bytes(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    t = _CRC_TABLE
    for b in data:
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _zeros_operator(nbytes):
    """32x32 GF(2) matrix (as 32 column words) that advances a raw CRC register over `nbytes` zero
    bytes: zlib's crc32_combine construction with the Castagnoli polynomial."""
    sq = lambda m: [_gf2_times(m, v) for v in m]
    m = sq(sq(sq([0x82F63B78] + [1 << i for i in range(31)])))     # one zero bit, cubed-squared: one byte
    op = [1 << i for i in range(32)]                                # identity
    while nbytes:
        if nbytes & 1:
            op = [_gf2_times(m, v) for v in op]                     # powers of one matrix commute
        m = sq(m)
        nbytes >>= 1
    return op


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C of `data`.  Long buffers are cut into equal chunks whose registers advance together
    as numpy vectors; the chunk CRCs are then chained with the zero-advance operator."""
    n = len(data)
    lanes = 2048
    if n < 64 * lanes:
        return _crc_bytes(data, crc)
    L = n // lanes
    body = np.frombuffer(data, np.uint8, lanes * L).reshape(lanes, L)
    c = np.full(lanes, 0xFFFFFFFF, np.uint32)
    t = _CRC_TABLE
    for j in range(L):
        c = t[(c ^ body[:, j]) & 0xFF] ^ (c >> np.uint32(8))
    c ^= np.uint32(0xFFFFFFFF)
    op = _zeros_operator(L)
    # crc(A || B) = advance(crc(A), len(B)) ^ crc(B) for the conditioned (init/xorout ~0) CRC
    total = _gf2_times(op, crc) ^ int(c[0]) if crc else int(c[0])
    for i in range(1, lanes):
        total = _gf2_times(op, total) ^ int(c[i])
    return _crc_bytes(data[lanes * L:], total)


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---- varints / minimal protobuf ------------------------------------------------------------------------
def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_fields(buf):
    """Yields (field_number, wire_type, value) of one serialized message."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            val = buf[pos:pos + n]; pos += n
        elif wt == 5:
            val = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, val


def _pb_tag(field, wt):
    return _put_varint((field << 3) | wt)


def _parse_entry(buf):
    """BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto)."""
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for f, wt, v in _pb_fields(buf):
        if f == 1: e["dtype"] = v
        elif f == 2:                                   # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
            for f2, _wt2, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _wt3, v3 in _pb_fields(v2):
                        if f3 == 1: size = v3
                    e["shape"].append(size)
                elif f2 == 3 and v2:
                    raise ValueError("tensor of unknown rank in checkpoint")
        elif f == 3: e["shard_id"] = v
        elif f == 4: e["offset"] = v
        elif f == 5: e["size"] = v
        elif f == 6: e["crc32c"] = struct.unpack("<I", v)[0]
        elif f == 7: e["slices"] += 1
    return e


def _encode_entry(dtype_id, shape, shard_id, offset, size, crc):
    dims = b"".join(_pb_tag(2, 2) + _put_varint(len(d)) + d
                    for d in (_pb_tag(1, 0) + _put_varint(int(s)) for s in shape))
    out = _pb_tag(1, 0) + _put_varint(dtype_id)
    out += _pb_tag(2, 2) + _put_varint(len(dims)) + dims
    if shard_id: out += _pb_tag(3, 0) + _put_varint(shard_id)
    if offset: out += _pb_tag(4, 0) + _put_varint(offset)
    out += _pb_tag(5, 0) + _put_varint(size)
    out += _pb_tag(6, 5) + struct.pack("<I", crc)
    return out


# ---- the table file -------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify):
    contents = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
        if mask_crc(crc32c(data[offset:offset + size + 1])) != stored:
            raise ValueError("index block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        raise NotImplementedError("snappy-compressed index blocks are not supported")
    if ctype != 0:
        raise ValueError("unknown block compression type %d" % ctype)
    return contents


def _block_entries(block):
    num_restarts = struct.unpack("<I", block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _read_table(path, verify=True):
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a tensor-bundle index (bad table magic)" % path)
    footer = data[-48:]
    _mo, pos = _get_varint(footer, 0)
    _ms, pos = _get_varint(footer, pos)
    io, pos = _get_varint(footer, pos)
    isz, pos = _get_varint(footer, pos)
    out = []
    for _sep, handle in _block_entries(_read_block(data, io, isz, verify)):
        bo, p = _get_varint(handle, 0)
        bs, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(data, bo, bs, verify)))
    return out


class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf = bytearray(); self.restarts = [0]; self.count = 0; self.last = b""; self.ri = restart_interval

    def add(self, key, value):
        shared = 0
        if self.count % self.ri == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key; self.count += 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _write_table(path, items, block_size=4096):
    out = bytearray()

    def emit(contents):
        off = len(out)
        out.extend(contents + b"\x00")
        out.extend(struct.pack("<I", mask_crc(crc32c(contents + b"\x00"))))
        return _put_varint(off) + _put_varint(len(contents))

    index = _BlockBuilder(restart_interval=1)
    blk = _BlockBuilder()
    for key, value in items:
        blk.add(key, value)
        if len(blk.buf) >= block_size:
            index.add(blk.last, emit(blk.finish()))
            blk = _BlockBuilder()
    if blk.count:
        index.add(blk.last, emit(blk.finish()))
    meta_handle = emit(_BlockBuilder().finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(bytes(out))


# ---- public API -----------------------------------------------------------------------------------------
def list_variables(prefix):
    """[(name, shape, numpy dtype)] of a checkpoint, like tf.train.list_variables."""
    out = []
    for key, value in _read_table(prefix + ".index"):
        if not key:
            continue
        e = _parse_entry(value)
        out.append((key.decode(), tuple(e["shape"]), _DTYPES.get(e["dtype"])))
    return out


def read_checkpoint(prefix, names=None, verify=True):
    """{variable name: ndarray} for all (or the named) variables of ``<prefix>.index`` + data shards."""
    entries, num_shards = {}, 1
    for key, value in _read_table(prefix + ".index", verify):
        if not key:                                    # BundleHeaderProto: num_shards = 1, endianness = 2
            for f, _wt, v in _pb_fields(value):
                if f == 1: num_shards = v
                elif f == 2 and v != 0:
                    raise NotImplementedError("big-endian checkpoint")
            continue
        entries[key.decode()] = _parse_entry(value)
    wanted = list(entries) if names is None else list(names)
    shards, out = {}, {}
    for name in wanted:
        if name not in entries:
            raise KeyError("variable %r not found in checkpoint %s" % (name, prefix))
        e = entries[name]
        if e["slices"]:
            raise NotImplementedError("partitioned variable %r" % name)
        if e["dtype"] not in _DTYPES:
            raise NotImplementedError("dtype %d of variable %r" % (e["dtype"], name))
        sid = e["shard_id"]
        if sid not in shards:
            with open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "rb") as f:
                shards[sid] = f.read()
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        dt = _DTYPES[e["dtype"]]
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if len(raw) != e["size"] or count * dt.itemsize != e["size"]:
            raise ValueError("variable %r: size %d does not match shape %s" % (name, e["size"], e["shape"]))
        if verify and e["crc32c"] is not None and mask_crc(crc32c(raw)) != e["crc32c"]:
            raise ValueError("variable %r: data checksum mismatch" % name)
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    return out


def write_checkpoint(prefix, tensors):
    """Writes {name: ndarray} as a one-shard V2 checkpoint (``<prefix>.index`` + ``.data-00000-of-00001``)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    header = _pb_tag(1, 0) + _put_varint(1) + _pb_tag(3, 2) + _put_varint(2) + _pb_tag(1, 0) + _put_varint(1)
    items, data = [(b"", header)], bytearray()
    for name in sorted(tensors, key=lambda s: s.encode()):
        a = np.asarray(tensors[name])                      # (ascontiguousarray would turn scalars into shape (1,))
        dt = a.dtype.newbyteorder("<")
        if dt not in _DTYPE_IDS:
            raise NotImplementedError("dtype %s" % a.dtype)
        raw = a.astype(dt).tobytes(order="C")
        items.append((name.encode(), _encode_entry(_DTYPE_IDS[dt], a.shape, 0, len(data), len(raw), mask_crc(crc32c(raw)))))
        data.extend(raw)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    _write_table(prefix + ".index", items)


def ckpt_path(base_dir, regularization, network_mode):
    """mvsnet/utils.py:75-90: <base_dir>/<regularization>/<network_mode>/model.ckpt"""
    return os.path.join(base_dir, regularization, network_mode, "model.ckpt")


def model_path(ckpt, ckpt_step):
    """mvsnet/utils.py:93-96"""
    return "-".join([ckpt, str(ckpt_step)])


# ---- variable-name map ----------------------------------------------------------------------------------
def variable_names(network_mode="normal", regularization="3DCNN", refinement=None):
    """{(group, layer, field): TensorFlow variable name} for the networks on the inference path.
    Names follow the layer names of mvsnet/cnn_wrapper/mvsnetworks.py:53-158 through
    tf.layers (``<layer>/kernel``), Network.batch_normalization (``<layer>/bn/{gamma,beta}``,
    network.py:278-298,492-509), group norm (``<layer>/gn/{gamma,beta}``, network.py:257-267),
    ConvGRUCell scopes (convgru.py:84-114; the three layer_norm calls per cell get the default
    scopes LayerNorm, LayerNorm_1 inside Gates and LayerNorm inside Output) and prob_conv
    (model.py:701-702)."""
    from .feature_net import UNET_LAYERS
    from .synthetic import make_regnet_params, gru_filters
    m = {}
    for name, kind, _srcs, _k, _mult, _stride in UNET_LAYERS:
        m[("unet", name, "w")] = name + "/kernel"
        if kind != "c":
            m[("unet", name, "gamma")] = name + "/gn/gamma"
            m[("unet", name, "beta")] = name + "/gn/beta"
    if regularization == "3DCNN":
        for name, p in make_regnet_params(network_mode).items():
            m[("regnet", name, "w")] = name + "/kernel"
            if "gamma" in p:
                m[("regnet", name, "gamma")] = name + "/bn/gamma"
                m[("regnet", name, "beta")] = name + "/bn/beta"
    elif regularization == "GRU":
        for i, _f in enumerate(gru_filters(network_mode), start=1):
            s = "conv_gru%d" % i
            g = "gru%d" % i
            m[("gru", g, "gates_w")] = s + "/Gates/conv/kernel"
            m[("gru", g, "gates_b")] = s + "/Gates/conv/bias"
            m[("gru", g, "out_w")] = s + "/Output/output_conv/kernel"
            m[("gru", g, "out_b")] = s + "/Output/output_conv/bias"
            for nm, scope in (("reset", "/Gates/LayerNorm"), ("update", "/Gates/LayerNorm_1"), ("out", "/Output/LayerNorm")):
                m[("gru", g, nm + "_gamma")] = s + scope + "/gamma"
                m[("gru", g, nm + "_beta")] = s + scope + "/beta"
        m[("gru", None, "prob_w")] = "prob_conv/kernel"
        m[("gru", None, "prob_b")] = "prob_conv/bias"
    else:
        raise NotImplementedError(regularization)
    if refinement:                                     # 'original' | 'unet' (mvsnetworks.py:178-193,261-324)
        from .refine import refine_layers
        for layer in refine_layers(refinement)[0]:
            m[("refine", layer[0], "w")] = layer[0] + "/kernel"
            m[("refine", layer[0], "b")] = layer[0] + "/bias"
    return m


def load_mvsnet_params(prefix, network_mode="normal", regularization="3DCNN", refinement=None):
    """Reads the inference-path variables of a reference checkpoint into this package's parameter
    dictionaries: returns {"unet": ..., "regnet": ... | None, "gru": ... | None} ready for
    ``MVSNetWeights.from_numpy``.  Missing variables raise KeyError naming the variable."""
    names = variable_names(network_mode, regularization, refinement)
    values = read_checkpoint(prefix, sorted(set(names.values())))
    out = {"unet": {}, "regnet": None, "gru": None, "refine": None}
    for (group, layer, field), var in names.items():
        if out[group] is None:
            out[group] = {}
        dst = out[group] if layer is None else out[group].setdefault(layer, {})
        dst[field] = values[var].astype(np.float32)
    return out


def export_mvsnet_params(prefix, unet=None, regnet=None, gru=None, network_mode="normal", refine=None,
                         refinement="original"):
    """Inverse of `load_mvsnet_params`: writes parameter dictionaries under the reference's names."""
    tensors = {}
    for reg, params in (("3DCNN", {"unet": unet, "regnet": regnet, "refine": refine}), ("GRU", {"unet": unet, "gru": gru})):
        for (group, layer, field), var in variable_names(network_mode, reg, refinement if refine is not None else None).items():
            src = params.get(group)
            if src is None:
                continue
            tensors[var] = np.asarray(src[field] if layer is None else src[layer][field], np.float32)
    write_checkpoint(prefix, tensors)
    return sorted(tensors)

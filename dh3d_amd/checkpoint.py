"""TensorFlow checkpoint (tensor bundle) -> DH3D state_dict, without TensorFlow.

The reference ships `models/{local,global}/*.index` (+ `.data-00000-of-00001` blobs, which its repository omits) and
restores them with `SaverRestore` / `get_model_loader` (`localdesc_extract.py:120-127`, `globaldesc_extract.py:85-91`,
`train.py:83-86`).  State-dict keys here are the checkpoint variable names with '/' -> '.' (model.tf_variable_name), so
importing is a table walk:

  `<prefix>.index`  an SSTable (LevelDB table format: prefix-compressed blocks + index block + 48-byte footer) mapping
                    variable name -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}; key "" holds the
                    BundleHeaderProto
  `<prefix>.data-SSSSS-of-NNNNN`  raw little-endian tensor bytes at [offset, offset+size)

Only what a saver writes is handled: uncompressed blocks, full (unsliced) tensors, float/double/int32/int64.
"""
import os
import struct

import numpy as np
import torch

from .model import tf_variable_name

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}  # tensorflow DataType enum


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def _block(buf, offset, size):
    """Entries of one table block: [(key bytes, value bytes)] (LevelDB block format, restart array ignored)."""
    if buf[offset + size] != 0:
        raise NotImplementedError("compressed table block (type %d)" % buf[offset + size])
    data = buf[offset:offset + size]
    n_restarts = struct.unpack_from("<I", data, size - 4)[0]
    end = size - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(data, pos)
        non_shared, pos = _varint(data, pos)
        vlen, pos = _varint(data, pos)
        key = key[:shared] + bytes(data[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(data[pos:pos + vlen])))
        pos += vlen
    return out


def _proto(buf):
    """Flat protobuf decode: {field number: [values]} (varints as int, length-delimited as bytes, fixed32 as int)."""
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _varint(buf, pos)
        elif wire == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wire == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("protobuf wire type %d" % wire)
        out.setdefault(field, []).append(v)
    return out


def read_index(index_path):
    """`<prefix>.index` -> (header dict, {variable name: {"dtype", "shape", "shard", "offset", "size", "crc32c"}})."""
    buf = open(index_path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise ValueError("%s is not a tensor-bundle index (bad table magic)" % index_path)
    pos = len(buf) - 48
    _, pos = _varint(buf, pos)  # metaindex handle
    _, pos = _varint(buf, pos)
    ioff, pos = _varint(buf, pos)
    isize, pos = _varint(buf, pos)
    header, entries = {}, {}
    for _, handle in _block(buf, ioff, isize):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for key, val in _block(buf, boff, bsize):
            msg = _proto(val)
            if key == b"":
                header = {"num_shards": msg.get(1, [1])[0], "endianness": msg.get(2, [0])[0]}
                continue
            if 7 in msg:
                raise NotImplementedError("sliced (partitioned) variable %r" % key.decode())
            shape = []
            for sh in msg.get(2, []):
                for dim in _proto(sh).get(2, []):
                    shape.append(_proto(dim).get(1, [0])[0])
            entries[key.decode()] = {"dtype": msg.get(1, [0])[0], "shape": tuple(shape), "shard": msg.get(3, [0])[0],
                                     "offset": msg.get(4, [0])[0], "size": msg.get(5, [0])[0],
                                     "crc32c": msg.get(6, [0])[0]}
    if header.get("endianness", 0) != 0:
        raise NotImplementedError("big-endian bundle")
    return header, entries


def read_checkpoint(prefix, names=None):
    """{variable name: numpy array} for `names` (default: every variable) of the bundle `<prefix>.index/.data-*`."""
    header, entries = read_index(prefix + ".index")
    shards, out = {}, {}
    for name in (entries if names is None else names):
        e = entries[name]
        if e["dtype"] not in _DTYPES:
            raise NotImplementedError("dtype enum %d of %r" % (e["dtype"], name))
        if e["shard"] not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, e["shard"], header.get("num_shards", 1))
            if not os.path.exists(path):
                raise FileNotFoundError("%s (the reference repository omits the weight blobs: "
                                        "models/.MISSING_LARGE_BLOBS)" % path)
            shards[e["shard"]] = np.memmap(path, dtype=np.uint8, mode="r")
        raw = shards[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        arr = np.frombuffer(bytes(raw), dtype=np.dtype(_DTYPES[e["dtype"]]).newbyteorder("<"))
        if arr.size != int(np.prod(e["shape"], dtype=np.int64)):
            raise ValueError("%r: %d bytes do not hold shape %s" % (name, e["size"], e["shape"]))
        out[name] = arr.reshape(e["shape"])
    return out


def load_tf_checkpoint(model, prefix, strict=True):
    """Fill `model` (a DH3D) from a TensorFlow checkpoint.  Returns (missing state_dict keys, unused variables);
    optimizer slots (`*/Adam`, `*/Adam_1`, `beta*_power`), `global_step`, `learning_rate` and EMA bookkeeping that has
    no module parameter are 'unused' by design.  strict: every state_dict key must be found with the right shape."""
    _, entries = read_index(prefix + ".index")
    sd = model.state_dict()
    wanted = {k: tf_variable_name(k) for k in sd}
    missing = [k for k, v in wanted.items() if v not in entries]
    if strict and missing:
        raise KeyError("checkpoint %s lacks %d variables, e.g. %s" % (prefix, len(missing), wanted[missing[0]]))
    found = {k: v for k, v in wanted.items() if v in entries}
    for k, v in found.items():
        if tuple(entries[v]["shape"]) != tuple(sd[k].shape):
            raise ValueError("%s: checkpoint shape %s, module shape %s" % (v, entries[v]["shape"], tuple(sd[k].shape)))
    arrays = read_checkpoint(prefix, names=sorted(set(found.values())))
    new = {k: torch.from_numpy(np.array(arrays[v])).to(dtype=sd[k].dtype) for k, v in found.items()}
    model.load_state_dict(new, strict=False)
    if hasattr(model, "invalidate"):
        model.invalidate()  # packed / folded copies of the weights are rebuilt on the next forward
    used = set(found.values())
    return missing, sorted(n for n in entries if n not in used)

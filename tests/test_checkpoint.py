"""TensorFlow tensor-bundle import (dh3d_amd/checkpoint.py).  Fixtures: the reference's own `models/*/*.index` tables
(data written by TensorFlow's saver; the weight blobs are absent upstream) + a bundle synthesised by the test-only writer
below, which follows the LevelDB table / BundleEntryProto layout independently of the reader."""
import os
import struct

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _entry_proto(dtype, shape, offset, size):
    dims = b"".join(b"\x12" + _vi(len(d)) + d for d in (b"\x08" + _vi(s) for s in shape))
    msg = b"\x08" + _vi(dtype) + b"\x12" + _vi(len(dims)) + dims
    if offset:
        msg += b"\x20" + _vi(offset)
    return msg + b"\x28" + _vi(size) + b"\x35" + struct.pack("<I", 0)


def _table_block(pairs, restart_every=16):
    """LevelDB block with prefix compression and a restart array."""
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i % restart_every == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(prev)) and k[shared] == prev[shared]:
                shared += 1
        out += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix, tensors, per_block=7):
    """Test-only tensor-bundle writer (no checksums): {name: numpy array} -> prefix.index + prefix.data-00000-of-00001."""
    names = sorted(tensors)
    dt = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}
    data, pairs = bytearray(), [(b"", b"\x08\x01\x1a\x02\x08\x01")]  # header: num_shards=1, version{producer=1}
    for n in names:
        a = np.ascontiguousarray(tensors[n])
        pairs.append((n.encode(), _entry_proto(dt[a.dtype], a.shape, len(data), a.nbytes)))
        data += a.tobytes()
    buf, index = bytearray(), []
    for i in range(0, len(pairs), per_block):
        blk = _table_block(pairs[i:i + per_block])
        index.append((pairs[min(i + per_block, len(pairs)) - 1][0] + b"\x00", _vi(len(buf)) + _vi(len(blk))))
        buf += blk + b"\x00" + struct.pack("<I", 0)  # trailer: no compression, crc (unchecked)
    meta_off = len(buf)
    meta = _table_block([])
    buf += meta + b"\x00" + struct.pack("<I", 0)
    idx_off = len(buf)
    idx = _table_block(index, restart_every=1)
    buf += idx + b"\x00" + struct.pack("<I", 0)
    footer = _vi(meta_off) + _vi(len(meta)) + _vi(idx_off) + _vi(len(idx))
    buf += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    open(prefix + ".index", "wb").write(buf)
    open(prefix + ".data-00000-of-00001", "wb").write(data)


@pytest.mark.parametrize("which,preset,nvars", [("globalmodel", "global_config", 155), ("localmodel", "detection_config", 133)])
def test_reads_the_reference_index_tables(which, preset, nvars):
    from dh3d_amd import ConfigFactory
    from dh3d_amd.checkpoint import read_index
    from dh3d_amd.model import DH3D, tf_variable_name
    header, entries = read_index(os.path.join(GOLD, which + ".index"))
    assert header["num_shards"] == 1 and len(entries) == nvars
    assert entries["global_step"]["dtype"] == 9 and entries["global_step"]["shape"] == ()
    # the data file is laid out in key order, back to back
    off = 0
    for name in sorted(entries):
        assert entries[name]["offset"] == off, name
        off += entries[name]["size"]
    # every module parameter / buffer is in the table with its shape; what is left over is optimizer + bookkeeping
    model = DH3D(ConfigFactory(preset).getconfig())
    used = set()
    for k, v in model.state_dict().items():
        name = tf_variable_name(k)
        assert name in entries, name
        assert entries[name]["shape"] == tuple(v.shape) and entries[name]["dtype"] == 1
        used.add(name)
    rest = [n for n in entries if n not in used]
    assert all(n.startswith("EMA/") or n.endswith(("/Adam", "/Adam_1")) or n in
               ("global_step", "learning_rate", "beta1_power", "beta2_power") for n in rest), rest


def test_missing_blob_is_reported():
    from dh3d_amd.checkpoint import read_checkpoint
    with pytest.raises(FileNotFoundError, match="data-00000-of-00001"):
        read_checkpoint(os.path.join(GOLD, "globalmodel"), names=["cluster_weights"])


def test_import_round_trip(tmp_path):
    from dh3d_amd import ConfigFactory
    from dh3d_amd.checkpoint import load_tf_checkpoint, read_checkpoint
    from dh3d_amd.model import DH3D, tf_variable_name
    cfg = ConfigFactory("global_config").getconfig()
    src = DH3D(cfg)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for t in src.state_dict().values():
            t.copy_(torch.randn(t.shape, generator=g) if t.dtype.is_floating_point else t)
    tensors = {tf_variable_name(k): v.numpy() for k, v in src.state_dict().items()}
    tensors["global_step"] = np.array(123, np.int64)                  # bookkeeping a saver adds
    tensors["cluster_weights/Adam"] = np.zeros((256, 64), np.float32)
    prefix = str(tmp_path / "model-123")
    write_bundle(prefix, tensors)
    back = read_checkpoint(prefix)
    assert set(back) == set(tensors) and int(back["global_step"].item()) == 123
    dst = DH3D(cfg)
    missing, unused = load_tf_checkpoint(dst, prefix)
    assert missing == [] and unused == ["cluster_weights/Adam", "global_step"]
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # a local-only checkpoint cannot fill the global head
    del_keys = [n for n in tensors if n.startswith(("cluster", "hidden1", "gating", "bn/", "globalatt", "global_before"))]
    write_bundle(prefix, {n: v for n, v in tensors.items() if n not in del_keys})
    with pytest.raises(KeyError):
        load_tf_checkpoint(DH3D(cfg), prefix)
    missing, _ = load_tf_checkpoint(DH3D(cfg), prefix, strict=False)
    assert sorted(tf_variable_name(k) for k in missing) == sorted(n for n in del_keys if not n.endswith("/Adam"))


@pytest.mark.gpu
def test_checkpoint_import_on_the_device_reproduces_the_descriptors(tmp_path):
    """The GPU leg of the import path (what localdesc_extract.py:120-127 / globaldesc_extract.py:85-91 do with
    SaverRestore): a TensorFlow-format bundle -> load_tf_checkpoint into a model that already lives on the GPU and has
    already run (stale packed weights must be rebuilt) -> same descriptors as the model the bundle was written from,
    and both equal to the oracle fed the bundle's tensors by their TensorFlow names."""
    from dh3d_amd import ConfigFactory
    from dh3d_amd.checkpoint import load_tf_checkpoint, read_checkpoint
    from dh3d_amd.model import DH3D, tf_variable_name
    from oracle import model_np
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    cfg = ConfigFactory("global_config").getconfig()
    src = DH3D(cfg).init_synthetic(12).to(dev).eval()
    pts = torch.rand(2, 1024, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        want = {k: v.clone() for k, v in src(pts).items() if k in ("xyz_feat", "globaldesc")}
    prefix = str(tmp_path / "model-7")
    write_bundle(prefix, {tf_variable_name(k): v.cpu().numpy() for k, v in src.state_dict().items()})
    dst = DH3D(cfg).init_synthetic(99).to(dev).eval()
    with torch.no_grad():
        before = dst(pts)["globaldesc"].clone()          # runs once with other weights: packed copies now exist
        missing, unused = load_tf_checkpoint(dst, prefix)
        got = dst(pts)
    assert missing == [] and unused == []
    assert not torch.allclose(before, want["globaldesc"])
    for k in want:
        assert torch.equal(got[k], want[k]), k
    exp = model_np.forward(pts.cpu().numpy(), read_checkpoint(prefix), extract_global=True)
    assert np.abs(got["globaldesc"].cpu().numpy() - exp["globaldesc"]).max() < 1e-4
    assert np.abs(got["xyz_feat"].cpu().numpy() - exp["xyz_feat"]).max() < 1e-4

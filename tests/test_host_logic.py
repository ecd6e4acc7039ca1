"""CPU: host-side logic -- configs, variable naming, losses, batch sharding, gloo all-gather (world 2)."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_presets():
    from dh3d_amd import ConfigFactory
    b = ConfigFactory("basic_config").getconfig()
    g = ConfigFactory("global_config").getconfig()
    assert b.num_points == 8192 and b.knn_num == 8 and b.featdim == 128 and not b.extract_global
    assert g.extract_global and g.gl_dilate == 8 and g.gl_dims == [256] and g.num_neg == 8 and g.other_neg
    assert g.some_missing_key is None  # dotdict semantics, core/configs.py:22-26


def test_state_dict_matches_checkpoint_variable_names():
    """Shapes from SURVEY 8(a)-0 (parsed from models/global/globalmodel.index)."""
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D, tf_variable_name
    m = DH3D(ConfigFactory("global_config").getconfig())
    sd = {tf_variable_name(k): tuple(v.shape) for k, v in m.state_dict().items()}
    expect = {
        "initconv/position_theta": (3, 32), "initconv/position_bias": (32,), "initconv_bn/mean/EMA": (32,),
        "stage1/flexconv_0/position_theta": (3, 32, 64), "stage1/flexconv_1/position_bias": (64, 64),
        "stage1/flexconv_0/feature_bias": (64, 1), "stage1/flexconv_1_bn/variance/EMA": (64,),
        "stage1/se/f1/tfconv0/W": (1, 1, 64, 16), "stage1/se/f2/tfconv0/b": (64,),
        "before_stage2_conv1d/tfconv0/W": (1, 1, 64, 64), "before_stage2_conv1d/tfconv0/bn/gamma": (64,),
        "stage2/flexconv_0/position_theta": (3, 64, 128), "stage2/flexconv_1/position_theta": (3, 128, 128),
        "stage2/se/f1/tfconv0/W": (1, 1, 128, 32), "stage2/concat_conv1d/tfconv0/W": (1, 1, 192, 128),
        "local_stage1_shortcut/tfconv0/W": (1, 1, 64, 128), "local_stage1_shortcut/tfconv0/bn/mean/EMA": (128,),
        "global_before_assemble/flexconv_0/position_theta": (3, 128, 256),
        "global_before_assemble/flexconv_0/feature_bias": (256, 1),
        "globalatt/detec_conv0/W": (1, 1, 256, 1024), "globalatt/detec_conv_fc/W": (1, 1, 1024, 1),
        "cluster_weights": (256, 64), "cluster_weights2": (1, 256, 64), "hidden1_weights": (16384, 256),
        "gating_weights": (256, 256), "cluster_bn/moving_mean": (64,), "bn/moving_variance": (256,),
        "gating_bn/beta": (256,),
    }
    for k, shp in expect.items():
        assert sd.get(k) == shp, (k, sd.get(k))
    n = sum(int(np.prod(s)) for s in sd.values())
    assert n == 4869553  # non-optimizer parameters of the global checkpoint minus 4 bookkeeping scalars
    d = DH3D(ConfigFactory("detection_config").getconfig())
    sdd = {tf_variable_name(k): tuple(v.shape) for k, v in d.state_dict().items()}
    assert sdd["detection_block_reliable/detec_conv2/W"] == (1, 1, 256, 1024)
    assert sdd["detection_block_reliable/detec_conv_fc/b"] == (1,)


def _quad_np(desc, B, P, Ng, m1=0.5, m2=0.2):
    D = desc.shape[1]
    q = desc[:B]; pos = desc[B:B + P * B].reshape(B, P, D); neg = desc[B + P * B:B + P * B + Ng * B].reshape(B, Ng, D)
    oth = desc[B + P * B + Ng * B:]
    best = ((pos - q[:, None]) ** 2).sum(2).min(1)
    trip = np.maximum(m1 + best[:, None] - ((neg - q[:, None]) ** 2).sum(2), 0).max(1).mean()
    sec = np.maximum(m2 + best[:, None] - ((neg - oth[:, None]) ** 2).sum(2), 0).max(1).mean()
    return trip, trip + sec


def test_losses_match_restatement():
    from dh3d_amd import losses
    rng = np.random.default_rng(0)
    B, P, Ng = 2, 2, 8
    desc = rng.standard_normal((B * (1 + P + Ng + 1), 16)).astype(np.float32)
    desc /= np.linalg.norm(desc, axis=1, keepdims=True)
    trip, quad = _quad_np(desc.astype(np.float64), B, P, Ng)
    t = torch.from_numpy(desc)
    assert abs(float(losses.lazy_quadruplet_loss(t, B, P, Ng)) - quad) < 1e-5
    assert abs(float(losses.lazy_triplet_loss(t[:B * (1 + P + Ng)], B, P, Ng)) - trip) < 1e-5


def test_shard_plan_covers_every_cloud_once():
    from dh3d_amd import dist as D
    for Bt, world in ((22, 8), (32, 8), (8, 1), (5, 2), (3, 8)):
        per, padded = D.shard_plan(Bt, world)
        assert padded >= Bt and padded == per * world
        seen = []
        for r in range(world):
            a, b = D.local_slice(Bt, r, world)
            seen += list(range(a, b))
        assert seen == list(range(Bt))
    pts = torch.arange(5 * 4 * 3, dtype=torch.float32).reshape(5, 4, 3)
    blk, mask = D.shard_batch(pts, 1, 2)
    assert blk.shape == (3, 4, 3) and mask.tolist() == [True, True, False]
    assert torch.equal(blk[:2], pts[3:5])


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
from dh3d_amd import dist as D
rank, world = D.init_from_env(backend="gloo")
Bt, Dd = 5, 8
full = torch.arange(Bt * Dd, dtype=torch.float32).reshape(Bt, Dd)
per, _ = D.shard_plan(Bt, world)
a, b = D.local_slice(Bt, rank, world)
local = torch.zeros(per, Dd); local[: b - a] = full[a:b]
out = D.all_gather_descriptors(local, Bt)
assert torch.equal(out, full), (rank, out)
t = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
assert t == float(world)
# differentiable all-gather: backward keeps this rank's slice of the gradient
from dh3d_amd.training import _AllGatherKeepOwn
x = torch.full((per, Dd), float(rank + 1), requires_grad=True)
g = _AllGatherKeepOwn.apply(x)
w = torch.arange(g.numel(), dtype=torch.float32).reshape(g.shape)
(g * w).sum().backward()
assert torch.equal(x.grad, w[rank * per:(rank + 1) * per]), (rank, x.grad)
D.barrier()
open(os.path.join(%r, "rank%%d.ok" %% rank), "w").write("ok")
"""


def test_all_gather_world_size_2_gloo(tmp_path):
    import socket
    with socket.socket() as sk:  # a free port: a stale rendezvous from an earlier run must not collide
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % (ROOT, str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()  # (stdout of 2 ranks interleaves)


def test_commuted_netvlad_training_algebra():
    """The rewrite planned for NetVLAD's training rows: sampled-row GEMMs + 64-wide per-point values
    reproduce autograd's V, asum and every gradient of the straightforward graph (float64, CPU)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "netvlad_commute_check.py")], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_zero_arena_hands_out_zeroed_independent_tensors():
    """pm.ZeroArena (the training step's accumulators): zeros without an arena, the recorded demand sizes the buffer at
    the next begin(), tensors are independent of the arena's version counter (not views), 256-byte spaced, cleared by
    begin(), and an exhausted arena falls back to torch.zeros."""
    import torch
    from dh3d_amd import pm
    cpu = torch.device("cpu")
    z = pm.zeros((3, 4), torch.float64, cpu)
    assert z.shape == (3, 4) and z.dtype == torch.float64 and float(z.abs().sum()) == 0.0
    a = pm.ZeroArena()
    with pm.zero_arena(a):
        a.begin(cpu)                                  # nothing known yet: every request falls back
        x = pm.zeros((3, 5), torch.float64, cpu)
        y = pm.zeros((7,), torch.float32, cpu)
        assert a.buf is None and float(x.sum()) == 0.0 and a.demand == 256 + 256
        a.begin(cpu)                                  # sized by the step before
        assert a.buf is not None and a.buf.numel() == 512
        x = pm.zeros((3, 5), torch.float64, cpu)
        y = pm.zeros((7,), torch.float32, cpu)
        assert not x._is_view() and not y._is_view()
        assert y.data_ptr() - x.data_ptr() == 256 and x.data_ptr() == a.buf.data_ptr()
        x += 1.0
        y += 2.0
        extra = pm.zeros((100,), torch.float32, cpu)  # beyond the buffer: a plain zeros tensor, demand recorded
        assert float(extra.sum()) == 0.0 and a.demand == 512 + 512
        v = x._version
        a.begin(cpu)                                  # grows to 1024 bytes; the old tensors keep the old storage
        assert a.buf.numel() == 1024 and x._version == v
        w = pm.zeros((3, 5), torch.float64, cpu)
        w += 3.0
        a.begin(cpu)
        assert float(w.sum()) == 0.0                  # same buffer this time: cleared by ONE fill
    assert pm.zeros((2,), torch.float32, cpu).data_ptr() != a.buf.data_ptr()   # outside the context: no arena


def test_frozen_scopes_leave_the_trainable_set():
    """backbone_scope(freeze) is freeze_variables(stop_gradient=False, skip_collection=True) (core/tf_utils.py:144-153):
    frozen variables are not handed to the optimiser (the gradient still flows through them)."""
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    from dh3d_amd.training import local_trainable_parameters
    cfg = ConfigFactory("detection_config").getconfig()
    m = DH3D(cfg).init_synthetic(0)
    det = {id(p) for p in m.detection_block_reliable.parameters()}
    both = {id(p) for p in local_trainable_parameters(m)}
    assert det and det <= both
    m.config.freezedetection = True
    assert not (det & {id(p) for p in local_trainable_parameters(m)})
    m.config.freezedetection, m.config.freezebackbone = False, True
    assert {id(p) for p in local_trainable_parameters(m)} == det


def test_weights_version_moves_with_every_kind_of_weight_change():
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    m = DH3D(ConfigFactory("global_config").getconfig()).init_synthetic(0)
    v0 = m.weights_version
    m.invalidate(head_only=True)
    v1 = m.weights_version
    m.mark_weights_changed()
    v2 = m.weights_version
    m.load_state_dict(m.state_dict())
    assert v0 < v1 < v2 < m.weights_version and m.__dict__.get("_bn_stale")


def test_unknown_fetch_name_is_an_error():
    """forward(fetch=...) computes only what is asked for: a misspelt name must not silently compute nothing."""
    import pytest
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    m = DH3D(ConfigFactory("global_config").getconfig()).init_synthetic(0)
    with pytest.raises(ValueError, match="unknown output name"):
        m.forward(torch.zeros(1, 64, 3), fetch=("global_desc",))
    assert "globaldesc" in DH3D.OUTPUT_NAMES and "xyz_feat" in DH3D.OUTPUT_NAMES

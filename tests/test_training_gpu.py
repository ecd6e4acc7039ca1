"""GPU: the sharded Siamese training step of the global stage (BASELINE config 4) -- differentiable head vs
the fused inference path, gradient consistency under sharding, and a few optimisation steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(dev, seed=0, B=1, P=2, Ng=3):
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    cfg = ConfigFactory("global_config").getconfig()
    cfg.batch_size, cfg.num_pos, cfg.num_neg = B, P, Ng
    m = DH3D(cfg).init_synthetic(seed)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, buf in m.named_buffers():
            if name.endswith(("mean_EMA", "moving_mean")):
                buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
            elif name.endswith(("variance_EMA", "moving_variance")):
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
    return m.to(dev).eval().prepare()


def test_autograd_head_matches_fused_path_in_eval_mode(dev):
    from torch_reference import global_head_autograd
    m = _build(dev)
    pts = torch.rand(3, 1024, 3, generator=torch.Generator().manual_seed(101)).to(dev)
    with torch.no_grad():
        outs = m(pts)
        geo = m._geometry(pts, None)
        _, local = m.compute_local(pts, _geo=geo)
        lv = geo.level(8, 8)
        g = global_head_autograd(m, pts, local, lv, bn_training=False)
        g = g / g.norm(dim=1, keepdim=True).clamp_min(1e-4)
    assert torch.allclose(g, outs["globaldesc"], rtol=1e-4, atol=1e-4)


def test_directional_gradient_of_the_head(dev):
    """Analytic gradient (custom-op backward kernels + torch autograd) vs a central finite difference of the
    loss along a random direction in parameter space (f32: 2% tolerance)."""
    from torch_reference import global_head_autograd
    from dh3d_amd.training import trainable_head_parameters
    m = _build(dev, seed=3)
    pts = torch.rand(7, 512, 3, generator=torch.Generator().manual_seed(102)).to(dev)  # B=1: 1 anchor + 2 pos + 3 neg + 1 other-neg
    with torch.no_grad():
        geo = m._geometry(pts, None)
        _, local = m.compute_local(pts, _geo=geo)
        lv = geo.level(8, 8)
    R = torch.randn(7, 256, generator=torch.Generator().manual_seed(9)).to(dev)

    def loss_fn():  # smooth functional of the descriptors (the hinge/max of the real loss defeats finite differences)
        d = global_head_autograd(m, pts, local, lv, bn_training=False)
        d = d * torch.rsqrt((d * d).sum(1, keepdim=True).clamp_min(1e-8))
        return (d * R).sum() + (d * d.roll(1, 0)).sum()

    # (parameter, relative tolerance): the flex_conv weights sit in front of BN+ReLU kinks, where a finite
    # difference is only indicative (their backward kernels are checked against the oracle in test_ops_gpu)
    named = [(m.globalatt.detec_conv0.W, 0.06), (m.cluster_weights2, 0.03), (m.hidden1_weights, 0.03),
             (m.gating_weights, 0.03)]
    grads = torch.autograd.grad(loss_fn(), [p for p, _ in named])
    gen = torch.Generator(device="cpu").manual_seed(0)
    for (p, tol), g in zip(named, grads):
        d = torch.randn(p.shape, generator=gen).to(dev)
        analytic = (g * d).sum().item()
        eps = 1e-3
        with torch.no_grad():
            p.add_(eps * d); lp = loss_fn().item()
            p.sub_(2 * eps * d); lm = loss_fn().item()
            p.add_(eps * d)
        numeric = (lp - lm) / (2 * eps)
        assert abs(analytic - numeric) <= tol * max(abs(numeric), abs(analytic)) + 2e-2, (tuple(p.shape), analytic, numeric)
    assert len(trainable_head_parameters(m)) > 10


def test_shard_partial_gradients_sum_to_full_gradient(dev):
    """Emulates the 4-rank partition in one process (eval-mode BN so shards do not couple through statistics):
    sum over shards of d loss / d theta (each through its own clouds) == unsharded gradient."""
    from torch_reference import global_head_autograd
    from dh3d_amd import dist as D, losses
    m = _build(dev, seed=5, B=1, P=2, Ng=3)
    pts = torch.rand(7, 512, 3, generator=torch.Generator().manual_seed(103)).to(dev)
    theta = m.global_before_assemble.flexconv_0.position_theta
    Wg = m.gating_weights

    def descs(block):
        with torch.no_grad():
            geo = m._geometry(block, None)
            _, local = m.compute_local(block, _geo=geo)
            lv = geo.level(8, 8)
        d = global_head_autograd(m, block, local, lv, bn_training=False)
        return d * torch.rsqrt((d * d).sum(1, keepdim=True).clamp_min(1e-8))

    full = descs(pts)
    loss = losses.lazy_quadruplet_loss(full, 1, 2, 3)
    g_full = torch.autograd.grad(loss, [theta, Wg])
    acc = [torch.zeros_like(theta), torch.zeros_like(Wg)]
    world = 4
    per, _ = D.shard_plan(7, world)
    with torch.no_grad():
        all_desc = full.detach()
    for r in range(world):
        a, b = D.local_slice(7, r, world)
        if b <= a:
            continue
        blk, mask = D.shard_batch(pts, r, world)
        d_loc = descs(blk)[: b - a]
        gathered = torch.cat([all_desc[:a], d_loc, all_desc[b:]], 0)  # own slice differentiable, rest constant
        lr = losses.lazy_quadruplet_loss(gathered, 1, 2, 3)
        g = torch.autograd.grad(lr, [theta, Wg], allow_unused=True)
        for t, gi in zip(acc, g):
            if gi is not None:
                t += gi
    for a_, f_ in zip(acc, g_full):
        assert torch.allclose(a_, f_, rtol=1e-3, atol=1e-5 * float(f_.abs().max()) + 1e-7)


def test_training_steps_reduce_the_loss(dev):
    from dh3d_amd.training import QuadrupletTrainer
    m = _build(dev, seed=7, B=1, P=2, Ng=3)
    tr = QuadrupletTrainer(m, start_lr=2e-3)
    pts = torch.rand(7, 512, 3, generator=torch.Generator().manual_seed(104)).to(dev)
    before = [p.detach().clone() for p in tr.params[:3]]
    ls = [tr.step(pts) for _ in range(6)]
    assert all(np.isfinite(ls)) and ls[-1] < ls[0], ls
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.params[:3]))
    m.prepare()
    with torch.no_grad():
        out = m(pts)["globaldesc"]
    assert torch.isfinite(out).all()


def test_factorised_flex_conv_grads_match_the_drop_in_op(dev):
    """Training-path flex_conv (fused forward + factorised backward) against ops.flex_convolution (reference
    formulation, atomics backward): same outputs and the same gradients for features, theta, bias."""
    from dh3d_amd import ops, pm
    from dh3d_amd.training import flex_conv_factorised
    g = torch.Generator().manual_seed(21)
    B, M, K, Din, Dout = 3, 300, 8, 64, 128
    xyz = torch.rand(B, M, 3, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, K)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev).requires_grad_(True)
    f1, t1, b1 = mk(B, M, Din), mk(3, Din, Dout, sc=Din ** -0.5), mk(Din, Dout, sc=(8 * Din) ** -0.5)
    f2, t2, b2 = [v.detach().clone().requires_grad_(True) for v in (f1, t1, b1)]
    wgt = torch.randn(B, M, Dout, generator=g).to(dev)
    y1 = flex_conv_factorised(f1, xyz, nbr, t1, b1)
    y2 = ops.flex_convolution(f2.transpose(1, 2).contiguous(), xyz.transpose(1, 2).contiguous(),
                              nbr.transpose(1, 2).contiguous(), t2, b2).transpose(1, 2)
    assert (y1 - y2).abs().max().item() < 1e-5 * y2.abs().max().item()
    (y1 * wgt).sum().backward()
    (y2 * wgt).sum().backward()
    for a, b, name in ((f1.grad, f2.grad, "features"), (t1.grad, t2.grad, "theta"), (b1.grad, b2.grad, "bias")):
        err = (a - b).abs().max().item() / b.abs().max().item()
        assert err < 2e-5, (name, err)


_WORLD2 = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
from dh3d_amd import ConfigFactory, dist as D
from dh3d_amd.model import DH3D
from dh3d_amd.training import QuadrupletTrainer

sync_bn = bool(int(sys.argv[1]))
dev = torch.device("cuda", 0)            # both ranks share the one GPU; collectives are host-staged over gloo
torch.cuda.set_device(dev)

def build():
    cfg = ConfigFactory("global_config").getconfig()
    cfg.batch_size, cfg.num_pos, cfg.num_neg = 1, 2, 3          # 1 + 2 + 3 + 1 = 7 clouds -> 4 + 3(+1 padding)
    return DH3D(cfg).init_synthetic(5).to(dev).eval()

pts = torch.from_numpy(np.random.default_rng(77).random((7, 1024, 3), dtype=np.float32)).to(dev)
# single-process reference (no process group yet): whole-batch statistics, as the reference's one GPU
ref = build()
tr = QuadrupletTrainer(ref, sync_bn=False)
tr.keep_grads = True
ref_losses = [tr.step(pts)]
ref_grads = tr.last_grads
ref_bufs = {k: v.detach().clone() for k, v in ref.named_buffers()}

rank, world = D.init_from_env(backend="gloo")
assert world == 2
m = build()
t2 = QuadrupletTrainer(m, sync_bn=sync_bn)
t2.keep_grads = True
losses = [t2.step(pts)]
grads = t2.last_grads
bufs1 = {k: v.detach().clone() for k, v in m.named_buffers()}
losses.append(t2.step(pts))
assert all(np.isfinite(losses)), losses
# every rank must end with the same parameters (same gathered loss, SUM-reduced gradients)
flat = torch.cat([p.detach().reshape(-1) for p in t2.params])
both = D.all_gather_rows(flat[None])
assert torch.equal(both[0], both[1]), "ranks diverged"
if sync_bn:
    # sync-BN + sharding == the single-process step (summation order differs: tolerance, not equality)
    assert abs(losses[0] - ref_losses[0]) < 1e-5, (losses, ref_losses)
    for g, q in zip(grads, ref_grads):   # the reduced gradient of the first step (parameters after Adam are sign-like)
        # f32 sums in another order through three chained batch norms (variance as E[x^2] - E[x]^2), atomics in the
        # flex_conv backward: a few 1e-3 of the largest entry; a sharding bug (padding in the statistics, a lost
        # slice) shows up at 1e-1
        tol = 5e-3 * float(q.abs().max()) + 1e-7
        assert float((g - q).abs().max()) <= tol, (float((g - q).abs().max()), float(q.abs().max()))
    for k, v in bufs1.items():
        assert torch.allclose(v, ref_bufs[k], rtol=1e-4, atol=1e-5), k
# sync_bn off: per-rank statistics over 4 / 3 clouds are a different (legal) normalisation -- nothing to compare with
# the whole-batch step beyond finiteness and the ranks agreeing with each other (checked above)
D.barrier()
open(os.path.join(%(out)r, "rank%%d_%%d.ok" %% (rank, int(sync_bn))), "w").write("ok")
"""


_ONE_RANK = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as tdist
from dh3d_amd import ConfigFactory, dist as D
from dh3d_amd.model import DH3D
from dh3d_amd.training import QuadrupletTrainer

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)

def build():
    cfg = ConfigFactory("global_config").getconfig()
    cfg.batch_size, cfg.num_pos, cfg.num_neg = 1, 2, 3
    return DH3D(cfg).init_synthetic(5).to(dev).eval()

pts = torch.from_numpy(np.random.default_rng(78).random((7, 1024, 3), dtype=np.float32)).to(dev)
ref = build()
tr = QuadrupletTrainer(ref, sync_bn=False)              # plain single-process step (graphed after three eager ones)
ref_losses = [tr.step(pts) for _ in range(6)]
assert tr._step_graphs, "the plain step was not captured"

tdist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)s", rank=0, world_size=1)
D.FORCE_COLLECTIVES = True
m = build()
t1 = QuadrupletTrainer(m, sync_bn=True)                 # the sharded code path: every collective issued, on RCCL
assert t1.graph_step, "the sharded step must be capturable on RCCL"
D.COLLECTIVE_CALLS[0] = 0
losses = [t1.step(pts)]
per_step = D.COLLECTIVE_CALLS[0]
losses += [float(t1.step(pts, sync=False)) for _ in range(5)]
assert t1._step_graphs, "the sharded step fell back to eager launches"
assert per_step == 12, per_step   # 5 forward + 5 backward BatchNorm statistics, descriptor all-gather, gradient arena
assert np.allclose(losses, ref_losses, rtol=2e-3, atol=2e-4), (losses, ref_losses)
for p, q in zip(t1.params, tr.params):                  # six Adam steps later the parameters still agree: Adam moves an
    d = (p.detach() - q.detach()).abs()                   # entry by ~lr * sign(g) per step, so an entry whose gradient is
    assert float(d.max()) <= 6 * 2 * 5e-4 + 1e-6, float(d.max())   # rounding noise may differ by 2 * lr per step -- few do
    assert float(d.mean()) <= 1e-4, float(d.mean())
for p in t1.params:                                     # .grad are views of the one arena that was all-reduced
    assert p.grad.untyped_storage().data_ptr() == t1._garena.untyped_storage().data_ptr()
tdist.destroy_process_group()
open(os.path.join(%(out)r, "one_rank.ok"), "w").write("ok")
"""


def test_trainer_sharded_path_on_a_one_rank_rccl_group(dev, tmp_path):
    """The step as a sharded run issues it (sync-BN all-reduces, descriptor all-gather, ONE all-reduce of the flat
    gradient arena), replayed from a hipGraph WITH its RCCL collectives, on a 1-rank nccl group: same losses and
    parameters as the plain single-process step; 12 collectives per step."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    script = tmp_path / "one_rank.py"
    script.write_text(_ONE_RANK % {"root": root, "out": str(tmp_path), "port": port})
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (tmp_path / "one_rank.ok").exists()


@pytest.mark.parametrize("sync_bn", [1, 0])
def test_trainer_step_world_size_2_matches_single_process(dev, tmp_path, sync_bn):
    """QuadrupletTrainer.step with two ranks (gloo, collectives staged through the host, both ranks on this GPU):
    shard + masked padding + descriptor all-gather + sync-BN + gradient all-reduce == the single-process step."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    script = tmp_path / "w2.py"
    script.write_text(_WORLD2 % {"root": root, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(sync_bn)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (tmp_path / ("rank0_%d.ok" % sync_bn)).exists() and (tmp_path / ("rank1_%d.ok" % sync_bn)).exists()


def test_trainer_survives_a_rank_with_only_padding_clouds(dev):
    """22 clouds over 12 or 16 ranks leave tail ranks with an all-False mask: statistics of zero rows must not turn
    into NaN (which the SUM all-reduce would spread) and must leave the running buffers alone."""
    from torch_reference import batch_norm_train as _batch_norm_train
    x = torch.randn(2, 16, 8, device=dev, requires_grad=True)
    mask = torch.zeros(2, dtype=torch.bool, device=dev)
    rm, rv = torch.zeros(8, device=dev), torch.ones(8, device=dev)
    y = _batch_norm_train(x, 2, torch.ones(8, device=dev), torch.zeros(8, device=dev), rm, rv, 1e-5, 0.9, False, mask)
    (y * 0.0).sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
    assert torch.equal(rm, torch.zeros(8, device=dev)) and torch.equal(rv, torch.ones(8, device=dev))


def test_bn_train_kernels_match_torch(dev):
    """train_ops.batch_norm_train (HIP statistics / apply / backward sums / backward apply) against torch autograd on
    the textbook formula, with and without ReLU and padding mask."""
    from dh3d_amd import backbones as bb, train_ops as T
    g = torch.Generator().manual_seed(3)
    for (clouds, rpc, C, relu, use_mask) in [(3, 700, 256, True, False), (4, 512, 64, False, True), (5, 1, 256, False, True),
                                             (2, 4096, 1024, True, False), (22, 1, 256, True, False),
                                             (40, 1, 320, True, True), (64, 1, 64, False, False)]:   # short: one launch
        R = clouds * rpc
        x = (torch.randn(R, C, generator=g) * 2 + 0.5).to(dev)
        mask = None
        if use_mask:
            mask = torch.ones(clouds, dtype=torch.bool, device=dev); mask[-1] = False
        bn = bb.TPBatchNorm(C).to(dev)
        with torch.no_grad():
            bn.gamma.copy_(0.5 + torch.rand(C, generator=g).to(dev)); bn.beta.copy_(torch.randn(C, generator=g).to(dev))
        dy = torch.randn(R, C, generator=g).to(dev)
        x1 = x.clone().requires_grad_()
        y = T.batch_norm_train(x1, bn, relu, False, mask, rpc)
        y.backward(dy)
        g1, b1 = bn.gamma.grad.clone(), bn.beta.grad.clone()
        rm1, rv1 = bn.mean_EMA.clone(), bn.variance_EMA.clone()
        # torch restatement in float64
        bn.gamma.grad = None; bn.beta.grad = None
        x2 = x.double().clone().requires_grad_()
        live = torch.ones(R, dtype=torch.bool, device=dev) if mask is None else mask.repeat_interleave(rpc)
        xs = x2[live]
        mean, var = xs.mean(0), xs.var(0, unbiased=False)
        gam, bet = bn.gamma.double().detach().requires_grad_(), bn.beta.double().detach().requires_grad_()
        yy = (x2 - mean) * torch.rsqrt(var + bn.eps) * gam + bet
        if relu:
            yy = torch.relu(yy)
        yy = yy * live[:, None]
        (yy * dy.double()).sum().backward()
        assert torch.allclose(y[live], yy[live].float(), rtol=1e-5, atol=1e-5)
        for a, b in ((x1.grad, x2.grad), (g1, gam.grad), (b1, bet.grad)):
            assert float((a.double() - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7
        assert torch.allclose(rm1, (0.1 * mean).float(), rtol=1e-5, atol=1e-6)
        # tensorpack's BatchNorm runs the fused kernel: the moving variance takes the Bessel-corrected batch variance
        n = float(live.sum())
        assert torch.allclose(rv1, (0.9 + 0.1 * var * (n / (n - 1.0))).float(), rtol=1e-5, atol=1e-6)
        if clouds == 22:  # the un-fused site (cluster_bn): biased variance in the moving average
            sb = bb.SlimBatchNorm(C, fused=False).to(dev)
            T.batch_norm_train(x.clone(), sb, False, False, mask, rpc)
            assert torch.allclose(sb.moving_variance, (0.999 + 0.001 * var).float(), rtol=1e-5, atol=1e-6)


def test_hip_head_matches_torch_head_forward_and_gradients(dev):
    """global_head_hip (hand-written kernels in both directions) == global_head_autograd (plain torch) on the same
    batch: descriptors, every trainable gradient, BatchNorm running buffers; with and without a padding cloud."""
    from torch_reference import global_head_autograd
    from dh3d_amd.training import global_head_hip, trainable_head_parameters
    for use_mask in (False, True):
        res = []
        for impl in (global_head_hip, global_head_autograd):
            m = _build(dev, seed=11)
            pts = torch.rand(4, 4096, 3, generator=torch.Generator().manual_seed(5)).to(dev)
            mask = None
            if use_mask:
                mask = torch.tensor([True, True, True, False], device=dev)
            with torch.no_grad():
                geo = m._geometry(pts, None)
                _, local = m.compute_local(pts, _geo=geo)
                lv = geo.level(8, 8)
            if impl is global_head_hip:
                desc = impl(m, pts, local.detach(), lv, sync_bn=False, mask=mask)
            else:
                desc = impl(m, pts, local.detach(), lv, bn_training=True, sync_bn=False, mask=mask)
            live = desc if mask is None else desc[mask]
            wgt = torch.randn(live.shape, generator=torch.Generator().manual_seed(9)).to(dev)
            (live * wgt).sum().backward()
            params = trainable_head_parameters(m)
            res.append((live.detach(), [p.grad.clone() for p in params], {k: v.clone() for k, v in m.named_buffers()},
                        [n for n, p in m.named_parameters() if any(p is q for q in params)]))
        (d0, g0, b0, names), (d1, g1, b1, _) = res
        assert float((d0 - d1).abs().max()) <= 1e-4 * float(d1.abs().max()), float((d0 - d1).abs().max())
        for n, a, b in zip(names, g0, g1):
            # (biases in front of a BatchNorm have an exactly-zero gradient in theory: both sides hold ~1e-8 of rounding
            #  noise there, hence the absolute floor)
            # (a single-element gradient -- the fc bias: a sum of 16k signed terms -- is compared relative to itself,
            #  not to a tensor's largest entry: cancellation leaves it ~3e-3 of relative rounding)
            # (5e-3 of the largest entry: f32 sums in another order through three chained batch norms and f32 atomics
            #  in the scatter / split-K kernels -- run-to-run jitter of the HIP side alone is ~1e-3)
            tol = (5e-3 if b.numel() > 1 else 1e-2) * float(b.abs().max()) + 1e-6
            assert float((a - b).abs().max()) <= tol, (n, float((a - b).abs().max()), float(b.abs().max()))
        for k in b0:
            assert torch.allclose(b0[k], b1[k], rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("N,use_mask", [(4096, False), (4096, True), (5000, False)])
def test_commuted_attention_head_matches_the_materialised_one(dev, N, use_mask):
    """train_ops.attention_head_commuted (conv commuted through three_interpolate, pre-activation never written,
    csrc/interp_train.hip) == train_ops.attention_head on the materialised up-sampled rows: attention weights, the
    gradients of every parameter and of the sampled features, BatchNorm running buffers.  Same math reassociated; the
    scatter and the statistics use f32 / f64 atomics (run-to-run jitter ~1e-6)."""
    from dh3d_amd import ops, pm, train_ops as T
    g = torch.Generator().manual_seed(N + use_mask)
    Bt, M, Cin = 3, N // 8, 256
    pts = torch.rand(Bt, N, 3, generator=g).to(dev)
    samp = ops.farthest_point_sample(M, pts)
    cxyz = torch.gather(pts, 1, samp.long()[:, :, None].expand(-1, -1, 3)).contiguous()
    d3, i3 = ops.three_nn(pts, cxyz)
    order = pm.spatial_sort(pts)[0]
    mask = torch.tensor([True, False, True], device=dev) if use_mask else None
    res = []
    for commuted in (True, False):
        m = _build(dev, seed=3)
        conv, fcw = m.globalatt.detec_conv0, m.globalatt.detec_conv_fc
        with torch.no_grad():
            conv.bn.gamma.copy_(0.5 + torch.rand(conv.cout, generator=torch.Generator().manual_seed(1)).to(dev))
            conv.bn.beta.copy_(0.3 * torch.randn(conv.cout, generator=torch.Generator().manual_seed(2)).to(dev))
        coarse = torch.randn(Bt, M, Cin, generator=torch.Generator().manual_seed(7)).to(dev).requires_grad_(True)
        if commuted:
            att = T.attention_head_commuted(coarse.reshape(Bt * M, Cin), conv, fcw.W, fcw.b, i3, d3, order, False, mask)
        else:
            dd = torch.clamp(d3, min=1e-10)
            w = (1.0 / dd) / (1.0 / dd).sum(2, keepdim=True)
            up = ops.three_interpolate(coarse, i3, w.contiguous())
            att = T.attention_head(up.reshape(Bt * N, Cin), conv, fcw.W, fcw.b, False, mask, N)
        wgt = torch.randn(Bt * N, generator=torch.Generator().manual_seed(9)).to(dev)
        if mask is not None:
            wgt = wgt * mask.repeat_interleave(N).float()
        (att * wgt).sum().backward()
        ps = [conv.W, conv.bn.gamma, conv.bn.beta, fcw.W, fcw.b]
        res.append((att.detach(), coarse.grad.clone(), [p.grad.clone() for p in ps],
                    (conv.bn.mean_EMA.clone(), conv.bn.variance_EMA.clone())))
    (a0, c0, g0, b0), (a1, c1, g1, b1) = res
    live = slice(None) if mask is None else mask.repeat_interleave(N)
    assert float((a0[live] - a1[live]).abs().max()) < 2e-6
    # The two sides compute the pre-activation with different roundings (interp of a GEMM vs GEMM of an interp, ~1e-6
    # apart): of the 12 M ReLU inputs a handful lie that close to zero and take the other branch, which moves the
    # gradient entries they feed by up to ~1 % of the largest one.  So: the bulk must agree tightly (norm-wise), single
    # entries within 2 %.
    def agree(x, y, name):
        rel = float((x - y).norm() / (y.norm() + 1e-30))
        assert rel <= 2e-3, (name, rel)
        assert float((x - y).abs().max()) <= 2e-2 * float(y.abs().max()) + 1e-7, (name, float((x - y).abs().max()))
    agree(c0, c1, "dcoarse")
    for name, x, y in zip(("W", "gamma", "beta", "wfc", "bfc"), g0, g1):
        agree(x, y, name)
    assert torch.allclose(b0[0], b1[0], rtol=1e-5, atol=1e-6) and torch.allclose(b0[1], b1[1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N,use_mask,scrambled", [(4096, False, False), (4096, True, False), (5000, False, False),
                                                  (4096, False, True)])
def test_commuted_netvlad_assignment_matches_the_materialised_one(dev, N, use_mask, scrambled):
    """train_ops.netvlad_assign_commuted (NetVLAD's rows commuted through three_interpolate, nothing 256 wide written
    for the fine points, csrc/netvlad_train.hip + interp_train.hip MODE 4) == train_ops.netvlad_assign on the
    materialised up-sampled rows (core/backbones.py:202-256): V, asum, the gradients of the sampled rows, the
    attention, the cluster weights and the cluster BatchNorm, its running buffers.  Same math reassociated;
    statistics and scatters use f64 / f32 atomics.  `scrambled`: neighbours without spatial coherence -- every block
    exceeds the slot table and takes the overflow paths."""
    from dh3d_amd import ops, pm, train_ops as T
    g = torch.Generator().manual_seed(N + use_mask + 2 * scrambled)
    Bt, M, Dm = 3, N // 8, 256
    pts = torch.rand(Bt, N, 3, generator=g).to(dev)
    samp = ops.farthest_point_sample(M, pts)
    cxyz = torch.gather(pts, 1, samp.long()[:, :, None].expand(-1, -1, 3)).contiguous()
    d3, i3 = ops.three_nn(pts, cxyz)
    if scrambled:
        i3 = torch.randint(0, M, (Bt, N, 3), generator=g, dtype=torch.int32).to(dev)
    order = pm.spatial_sort(pts)[0]
    mask = torch.tensor([True, False, True], device=dev) if use_mask else None
    dV = torch.randn(Bt, 64, Dm, generator=torch.Generator().manual_seed(11)).to(dev)
    dA = torch.randn(Bt, 64, generator=torch.Generator().manual_seed(12)).to(dev)
    if mask is not None:   # the loss never sees a padding cloud's descriptor
        dV, dA = dV * mask[:, None, None].float(), dA * mask[:, None].float()
    res = []
    for commuted in (True, False):
        m = _build(dev, seed=3)
        nv = m._netvlad
        with torch.no_grad():
            nv.cluster_bn.gamma.copy_(0.5 + torch.rand(64, generator=torch.Generator().manual_seed(1)).to(dev))
            nv.cluster_bn.beta.copy_(0.3 * torch.randn(64, generator=torch.Generator().manual_seed(2)).to(dev))
        coarse = torch.randn(Bt, M, Dm, generator=torch.Generator().manual_seed(7)).to(dev).requires_grad_(True)
        att = torch.rand(Bt * N, generator=torch.Generator().manual_seed(8)).to(dev).requires_grad_(True)
        if commuted:
            V, asum = T.netvlad_assign_commuted(coarse, att, nv.cluster_weights, nv.cluster_bn, i3, d3, order, False, mask)
        else:
            dd = torch.clamp(d3, min=1e-10)
            w = (1.0 / dd) / (1.0 / dd).sum(2, keepdim=True)
            up = ops.three_interpolate(coarse, i3, w.contiguous())
            V, asum = T.netvlad_assign(up, att, nv.cluster_weights, nv.cluster_bn, False, mask)
        ((V * dV).sum() + (asum * dA).sum()).backward()
        ps = [nv.cluster_weights, nv.cluster_bn.gamma, nv.cluster_bn.beta]
        res.append((V.detach(), asum.detach(), coarse.grad.clone(), att.grad.clone(), [p.grad.clone() for p in ps],
                    (nv.cluster_bn.moving_mean.clone(), nv.cluster_bn.moving_variance.clone())))
    (V0, A0, c0, t0, g0, b0), (V1, A1, c1, t1, g1, b1) = res
    live = slice(None) if mask is None else mask
    def agree(x, y, name, tol=2e-4):
        rel = float((x - y).norm() / (y.norm() + 1e-30))
        assert rel <= tol, (name, rel)
        assert float((x - y).abs().max()) <= 10 * tol * float(y.abs().max()) + 1e-7, (name, float((x - y).abs().max()))
    agree(V0[live], V1[live], "V", 2e-5)
    agree(A0[live], A1[live], "asum", 2e-5)
    agree(c0, c1, "dcoarse")
    agree(t0, t1, "datt")
    for name, x, y in zip(("Wc", "gamma", "beta"), g0, g1):
        agree(x, y, name)
    assert torch.allclose(b0[0], b1[0], rtol=1e-5, atol=1e-6) and torch.allclose(b0[1], b1[1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,N,scrambled", [(3, 4096, False), (2, 5000, False), (2, 4096, True)])
def test_three_interpolate_sorted_backward_matches_the_drop_in_op(dev, B, N, scrambled):
    """train_ops.three_interpolate_sorted: same forward kernel, backward on the Morton order (MFMA scatter in LDS, one
    atomic per block, row and channel) == ops.three_interpolate's backward (threeinterpolate_grad_cpu,
    tf_interpolate.cpp:131-153) up to the summation order.  `scrambled`: neighbours without spatial coherence -- every
    block exceeds the 56 staged rows and takes the overflow path."""
    from dh3d_amd import ops, pm, train_ops as T
    g = torch.Generator().manual_seed(N + B)
    M, C = N // 8, 256
    pts = torch.rand(B, N, 3, generator=g).to(dev)
    samp = ops.farthest_point_sample(M, pts)
    cxyz = torch.gather(pts, 1, samp.long()[:, :, None].expand(-1, -1, 3)).contiguous()
    d3, i3 = ops.three_nn(pts, cxyz)
    if scrambled:
        i3 = torch.randint(0, M, (B, N, 3), generator=g, dtype=torch.int32).to(dev)
    dd = torch.clamp(d3, min=1e-10)
    w = ((1.0 / dd) / (1.0 / dd).sum(2, keepdim=True)).contiguous()
    order = pm.spatial_sort(pts)[0]
    go = torch.randn(B, N, C, generator=g).to(dev)
    outs = []
    for fn in (lambda p: T.three_interpolate_sorted(p, i3, w, order), lambda p: ops.three_interpolate(p, i3, w)):
        p = torch.randn(B, M, C, generator=torch.Generator().manual_seed(4)).to(dev).requires_grad_(True)
        y = fn(p)
        y.backward(go)
        outs.append((y.detach(), p.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    err = float((outs[0][1] - outs[1][1]).abs().max())
    assert err <= 2e-5 * float(outs[1][1].abs().max()), err


def test_trainer_backbone_graph_matches_eager_and_survives_a_reload(dev):
    """QuadrupletTrainer replays the frozen backbone + geometry from a hipGraph: same losses and gradients as the eager
    backbone, for two different batches (the graph's input buffer is refilled); after load_state_dict the graph is
    re-captured (it would otherwise read the freed packed weights of the old model)."""
    from dh3d_amd.training import QuadrupletTrainer
    batches = [torch.rand(7, 1024, 3, generator=torch.Generator().manual_seed(s)).to(dev) for s in (1, 2)]
    res = []
    for graph in (True, False):
        m = _build(dev, seed=21, B=1, P=2, Ng=3)
        tr = QuadrupletTrainer(m, start_lr=1e-3, graph_backbone=graph)
        tr.keep_grads = True
        out = []
        for b in batches:
            loss = tr.step(b)
            out.append((loss, [g.clone() for g in tr.last_grads]))
        res.append((out, tr, m))
    for (la, ga), (lb, gb) in zip(res[0][0], res[1][0]):
        # (the second step sees parameters moved by Adam's sign-like first update: the ~1e-6 run-to-run jitter of the
        #  atomics-summed gradients shows up as ~1e-5 in its loss)
        assert abs(la - lb) <= 1e-4 * max(1.0, abs(lb)), (la, lb)
        for x, y in zip(ga, gb):
            assert float((x - y).abs().max()) <= 5e-3 * float(y.abs().max()) + 1e-6
    # reload other backbone weights into the graphed trainer's model: the next step must see them
    tr, m = res[0][1], res[0][2]
    assert len(tr._bb_graphs) == 1
    other = _build(dev, seed=99, B=1, P=2, Ng=3)
    m.load_state_dict(other.state_dict())
    tr2 = QuadrupletTrainer(other, start_lr=1e-3, graph_backbone=False)
    la, lb = tr.step(batches[0]), tr2.step(batches[0])
    assert abs(la - lb) <= 1e-4 * max(1.0, abs(lb)), (la, lb)
    assert len(tr._bb_graphs) == 1  # the stale graph was dropped, a new one captured


def test_trainer_whole_step_graph_matches_eager_steps(dev):
    """graph_step=True (forward + loss + backward + weight decay + Adam captured into one hipGraph after three eager
    steps, replayed with the batch copied into its input buffer) follows the same trajectory as eager steps: losses
    over eight steps on alternating batches, the learning-rate staircase included (decay_step = 4)."""
    from dh3d_amd.training import QuadrupletTrainer
    batches = [torch.rand(7, 1024, 3, generator=torch.Generator().manual_seed(s)).to(dev) for s in (11, 12)]
    traj = []
    for graph in (True, False):
        m = _build(dev, seed=31, B=1, P=2, Ng=3)
        tr = QuadrupletTrainer(m, start_lr=5e-4, decay_step=4, decay_rate=0.5, graph_step=graph, graph_backbone=graph)
        assert tr.graph_step == graph
        ls = [tr.step(batches[i % 2]) for i in range(8)]
        traj.append((ls, [p.detach().clone() for p in tr.params], tr))
    (la, pa, tra), (lb, pb, _) = traj
    assert len(tra._step_graphs) == 1
    assert all(np.isfinite(la)) and all(np.isfinite(lb))
    # Adam's sign-like first steps amplify rounding differences of the (atomics-summed) gradients: the trajectories
    # agree to a few 1e-3 in the loss, not bit for bit
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-2 * max(1.0, abs(y)), (la, lb)
    drift = max(float((x - y).abs().max()) for x, y in zip(pa, pb))
    assert drift <= 5e-3, drift


def test_trainer_zero_arena_and_input_buffer(dev, monkeypatch):
    """(a) The step's accumulators come from ONE zeroed arena (pm.ZeroArena): same gradients as with every accumulator a
    torch.zeros of its own, and the arena is in use (non-empty, its demand stable from step to step).  (b) A batch
    written straight into the replayed step's input buffer (QuadrupletTrainer.input_buffer) gives the step it gives
    when passed as a tensor of its own (which is copied into that buffer)."""
    from dh3d_amd import pm
    from dh3d_amd.training import QuadrupletTrainer
    batches = [torch.rand(7, 4096, 3, generator=torch.Generator().manual_seed(s)).to(dev) for s in (41, 42)]
    grads = []
    for arena in (True, False):
        if not arena:   # every request falls back to torch.zeros
            monkeypatch.setattr(pm.ZeroArena, "take", lambda self, shape, dtype, device: torch.zeros(shape, dtype=dtype, device=device))
        m = _build(dev, seed=51, B=1, P=2, Ng=3)
        # (a vanishing rate: Adam's first update moves EVERY parameter by +-lr whatever its gradient's size, so the
        #  run-to-run rounding of the atomics-summed gradients flips signs where |g| ~ 0 and two identical trainers'
        #  SECOND-step gradients differ by ~1 % at lr = 1e-3 -- tools/arena_diag.py; the arena is what is compared here)
        tr = QuadrupletTrainer(m, start_lr=1e-8, graph_step=False)
        tr.keep_grads = True
        out = []
        for b in batches:
            loss = tr.step(b)
            out.append((loss, [g.clone() for g in tr.last_grads]))
        if arena:
            assert tr._zarena.buf is not None and tr._zarena.buf.numel() > 1 << 20 and tr._zarena.off > 0
            assert tr._zarena.demand == tr._zarena.peak   # the same requests every step
        grads.append(out)
    monkeypatch.undo()
    for (la, ga), (lb, gb) in zip(*grads):
        assert abs(la - lb) <= 1e-4 * max(1.0, abs(lb)), (la, lb)
        for x, y in zip(ga, gb):
            assert float((x - y).abs().max()) <= 5e-3 * float(y.abs().max()) + 1e-6
    # (b) two graph-replaying trainers on the same trajectory; one is fed through its input buffer
    losses = []
    for zero_copy in (True, False):
        m = _build(dev, seed=61, B=1, P=2, Ng=3)
        tr = QuadrupletTrainer(m, start_lr=5e-4)
        ls = [tr.step(batches[0]) for _ in range(4)]          # three eager steps, then the capture
        assert tr.input_buffer(batches[0].shape) is not None and tr.input_buffer((3, 5, 3)) is None
        for i in range(4):
            nxt = batches[(i + 1) % 2]
            if zero_copy:
                buf = tr.input_buffer(nxt.shape)
                buf.copy_(nxt)                                   # the loader's write
                ls.append(tr.step(buf))
            else:
                ls.append(tr.step(nxt))
        losses.append(ls)
    for x, y in zip(*losses):
        assert abs(x - y) <= 2e-2 * max(1.0, abs(y)), losses


def test_trainer_graphs_of_two_shapes_keep_their_own_arenas(dev):
    """A small batch shape is captured, then a LARGER one runs its eager warm-up (the eager accumulator arena grows
    and frees its old buffer) and is captured too; steps on the two shapes then alternate.  Every captured graph owns
    a fixed arena (pm.ZeroArena(fixed_bytes=...)), so the small shape's replays never touch freed memory, both
    graphs stay alive (no re-capture per switch), and the trajectory follows an eager trainer's."""
    from dh3d_amd.training import QuadrupletTrainer
    small = [torch.rand(7, 1024, 3, generator=torch.Generator().manual_seed(s)).to(dev) for s in (71, 72)]
    large = [torch.rand(7, 4096, 3, generator=torch.Generator().manual_seed(s)).to(dev) for s in (73, 74)]
    order = [small[0]] * 4 + [large[0]] * 4 + [small[1], large[1], small[0], large[0], small[1], large[1]]
    traj = []
    for graph in (True, False):
        m = _build(dev, seed=81, B=1, P=2, Ng=3)
        tr = QuadrupletTrainer(m, start_lr=5e-4, graph_step=graph, graph_backbone=graph)
        ls = []
        for i, b in enumerate(order):
            ls.append(tr.step(b))
            if graph and i >= 8:
                # scribble over whatever the caching allocator hands out now: a replay into a freed arena would
                # accumulate onto this instead of onto zeros
                junk = torch.full((1 << 22,), 1e30, device=dev)
                del junk
        traj.append((ls, tr))
    (la, tra), (lb, _) = traj
    assert len(tra._step_graphs) == 2, list(tra._step_graphs)
    arenas = [ent[3] for ent in tra._step_graphs.values()]
    assert all(a.fixed and a.buf is not None for a in arenas)
    assert arenas[0].buf.data_ptr() != arenas[1].buf.data_ptr()
    assert all(np.isfinite(la)) and all(np.isfinite(lb))
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-2 * max(1.0, abs(y)), (la, lb)


def test_vlad_normalize_kernels_match_the_tensor_expression(dev):
    """train_ops.vlad_normalize (one launch per direction) == V^T - asum*W2 -> intra-normalise -> flatten -> L2-normalise
    written with tensor ops (core/backbones.py:241-262), values and all three gradients, in float64 on the torch side."""
    from dh3d_amd import train_ops as T
    g = torch.Generator().manual_seed(3)
    Bt, Cl, Dm = 5, 64, 256
    V0 = torch.randn(Bt, Cl, Dm, generator=g).to(dev); a0 = torch.rand(Bt, Cl, generator=g).to(dev)
    W0 = (torch.randn(1, Dm, Cl, generator=g) / 16).to(dev)
    go = torch.randn(Bt, Dm * Cl, generator=g).to(dev)
    V, a, W = [t.clone().requires_grad_(True) for t in (V0, a0, W0)]
    y = T.vlad_normalize(V, a, W)
    y.backward(go)
    Vd, ad, Wd = [t.double().clone().requires_grad_(True) for t in (V0, a0, W0)]
    r = Vd.transpose(1, 2) - ad.unsqueeze(1) * Wd
    r = r * torch.rsqrt(torch.clamp((r * r).sum(1, keepdim=True), min=1e-12))
    r = r.reshape(Bt, Dm * Cl)
    r = r * torch.rsqrt(torch.clamp((r * r).sum(1, keepdim=True), min=1e-12))
    r.backward(go.double())
    assert float((y.double() - r).abs().max()) < 1e-6
    for name, x, ref in (("dV", V.grad, Vd.grad), ("dasum", a.grad, ad.grad), ("dW2", W.grad, Wd.grad)):
        assert float((x.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-9, name


@pytest.mark.parametrize("B,P,Ng,scale", [(1, 2, 18, 1.0), (3, 2, 5, 1.0), (2, 3, 7, 0.05), (1, 2, 18, 3.0)])
def test_quadruplet_loss_kernel_matches_the_tensor_expression(dev, B, P, Ng, scale):
    """train_ops.quadruplet_loss (loss and gradient in one launch) == losses.lazy_quadruplet_loss under autograd
    (core/losses.py:137-200): active and inactive hinges (`scale` moves the descriptors apart / together)."""
    from dh3d_amd import losses, train_ops as T
    g = torch.Generator().manual_seed(B * 100 + Ng)
    d0 = torch.nn.functional.normalize(torch.randn(B * (2 + P + Ng), 256, generator=g), dim=1).to(dev) * scale
    a = d0.clone().requires_grad_(True)
    la = T.quadruplet_loss(a, B, P, Ng, 0.5, 0.2)
    (la * 1.7).backward()
    b = d0.clone().requires_grad_(True)
    lb = losses.lazy_quadruplet_loss(b, B, P, Ng, 0.5, 0.2)
    (lb * 1.7).backward()
    assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(lb)))
    assert float((a.grad - b.grad).abs().max()) <= 1e-6 * max(1.0, float(b.grad.abs().max()))


def test_small_fused_training_ops(dev):
    """The one-launch forms of three groups of tiny tensor ops: inverse-distance weights, NetVLAD's per-cloud assignment
    sums from the assignment pass, context gating (forward and backward) -- against the torch expressions they replace."""
    from dh3d_amd import pm, train_ops as T
    g = torch.Generator().manual_seed(9)
    dist = torch.rand(3, 700, 3, generator=g).to(dev) * 0.1
    dist[0, :5, 0] = 0.0                                           # coincident points: clamp at 1e-10
    d = torch.clamp(dist, min=1e-10)
    ref = (1.0 / d) / (1.0 / d).sum(2, keepdim=True)
    assert torch.allclose(pm.idw_weights(dist), ref, rtol=2e-6, atol=1e-12)
    # assignment rows + per-cloud sums
    Bt, N = 3, 640
    s = torch.randn(Bt * N, 64, generator=g).to(dev) * 3
    scale, shift = (0.5 + torch.rand(64, generator=g)).to(dev), torch.randn(64, generator=g).to(dev)
    att = torch.rand(Bt * N, generator=g).to(dev)
    a_ref = torch.softmax(s.double() * scale.double() + shift.double(), dim=1) * att.double()[:, None]
    a1 = pm.netvlad_assign_rows(s, scale, shift, att)
    a2, asum = pm.netvlad_assign_rows(s, scale, shift, att, rows_per_cloud=N)
    assert torch.equal(a1, a2)
    assert float((a1.double() - a_ref).abs().max()) < 1e-6
    assert float((asum.double() - a_ref.reshape(Bt, N, 64).sum(1)).abs().max()) < 1e-4
    # context gate
    v = torch.randn(22, 256, generator=g).to(dev).requires_grad_()
    gt = (torch.randn(22, 256, generator=g) * 3).to(dev).requires_grad_()
    dy = torch.randn(22, 256, generator=g).to(dev)
    y = T.context_gate(v, gt)
    y.backward(dy)
    v2, g2 = v.detach().double().requires_grad_(), gt.detach().double().requires_grad_()
    y2 = v2 * torch.sigmoid(g2)
    y2.backward(dy.double())
    assert float((y.double() - y2).abs().max()) < 1e-6
    assert float((v.grad.double() - v2.grad).abs().max()) < 1e-6 and float((gt.grad.double() - g2.grad).abs().max()) < 1e-6


@pytest.mark.parametrize("variant", ["conv1d", "concat_xyz"])
def test_hip_head_trains_the_other_global_front_ends(dev, variant):
    """global_backbone='global_before_assemble_conv1d' (core/backbones.py:189-197) and concat_xyz=True (:180-181) under
    the HIP training head: forward, gradients of every trainable parameter and running statistics against the torch
    head (which runs concat_xyz literally -- 131 input channels through the drop-in flex_convolution)."""
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    from torch_reference import global_head_autograd
    from dh3d_amd.training import global_head_hip, trainable_head_parameters, QuadrupletTrainer
    res = []
    for impl in (global_head_hip, global_head_autograd):
        cfg = ConfigFactory("global_config").getconfig()
        cfg.batch_size, cfg.num_pos, cfg.num_neg = 1, 2, 3
        if variant == "conv1d":
            cfg.global_backbone = "global_before_assemble_conv1d"
        else:
            cfg.concat_xyz = True
        m = DH3D(cfg).init_synthetic(21).to(dev).eval().prepare()
        pts = torch.rand(3, 1024, 3, generator=torch.Generator().manual_seed(6)).to(dev)
        with torch.no_grad():
            geo = m._geometry(pts, None)
            _, local = m.compute_local(pts, _geo=geo)
            lv = geo.level(8, 8)
            m._join_side(geo)
        if impl is global_head_hip:
            desc = impl(m, pts, local.detach(), lv, sync_bn=False, mask=None)
        else:
            desc = impl(m, pts, local.detach(), lv, bn_training=True, sync_bn=False, mask=None)
        wgt = torch.randn(desc.shape, generator=torch.Generator().manual_seed(9)).to(dev)
        (desc * wgt).sum().backward()
        params = trainable_head_parameters(m)
        assert all(p.grad is not None for p in params)
        byid = {id(p): n for n, p in m.named_parameters()}
        res.append((desc.detach(), [p.grad.clone() for p in params], {k: v.clone() for k, v in m.named_buffers()},
                    [byid[id(p)] for p in params]))
    (d0, g0, b0, names), (d1, g1, b1, _) = res
    assert float((d0 - d1).abs().max()) <= 1e-4 * float(d1.abs().max())
    gmax = max(float(b.abs().max()) for b in g1)
    for n, a, b in zip(names, g0, g1):
        err, ref = float((a - b).abs().max()), float(b.abs().max())
        if n.endswith(("feature_bias", ".b")) and not n.endswith("detec_conv_fc.b"):
            # a bias in front of a BatchNorm: its gradient is exactly zero in theory, both sides hold rounding noise of
            # their own summation order there (column sums of a batch-norm backward)
            assert float(a.abs().max()) <= 1e-3 * gmax and ref <= 1e-3 * gmax, (n, float(a.abs().max()), ref, gmax)
            continue
        # (the two sides round the front end differently -- concat_xyz: one 131-channel convolution against the split
        #  factorised pair -- and three chained batch norms over a few hundred rows amplify that: entries within 5 % of
        #  the largest one, the tensor within 2 % in norm; a wrong slice or a missing term would be O(1))
        nerr = float((a - b).norm()) / (float(b.norm()) + 1e-12)
        assert err <= 5e-2 * ref + 1e-6 and nerr <= 2e-2, (n, err, ref, nerr)
    for k in b0:
        assert torch.allclose(b0[k], b1[k], rtol=1e-4, atol=1e-5), k
    # and a whole trainer step runs on it
    cfg.num_points = 1024
    tr = QuadrupletTrainer(m, graph_step=False)
    batch = torch.rand(1 + 2 + 3 + 1, 1024, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    l0 = tr.step(batch)
    assert l0 == l0 and l0 >= 0.0

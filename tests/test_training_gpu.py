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
    from dh3d_amd.training import global_head_autograd
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
    from dh3d_amd.training import global_head_autograd, trainable_head_parameters
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
    from dh3d_amd.training import global_head_autograd
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

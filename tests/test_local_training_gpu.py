"""GPU: the stage 1-2 training step (dh3d_amd.training.LocalTrainer; core/model.py:135-246 with basic_config /
detection_config, core/losses.py:29-133) -- the WHOLE local backbone (and the detector) in training mode, forward and
backward on HIP kernels.

  * forward against the oracle's training-mode graph (oracle/model_np.local_training_step_forward): loss, descriptors,
    every moving average the step updates, at 2 x (anchor + positive) x 4096 points;
  * gradients of every trainable tensor against a float64 torch restatement of the same graph (gather-based flex_conv /
    conv_pointset / flex_pool written with tensor ops, autograd's gradients);
  * the captured whole-step hipGraph follows eager steps; a short run reduces the loss.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PAIRS, N, M = 2, 4096, 256
# f32 kernels (bf16x6 / exact-f32 GEMMs, f32 atomics in the scatters) against a float64 graph through ~20 layers with ten
# BatchNorm backward passes: measured worst 3.2e-3 (stage2 position_theta, a tensor 50x smaller than the largest gradient)
TOL_GRAD = 6e-3


def _weights_np(model):
    from dh3d_amd.model import tf_variable_name
    return {tf_variable_name(k): v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def _build(dev, preset, seed=5):
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    cfg = ConfigFactory(preset).getconfig()
    cfg.num_points, cfg.batch_size, cfg.sampled_kpnum = N, PAIRS, M
    m = DH3D(cfg).init_synthetic(seed)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, buf in m.named_buffers():
            if name.endswith("mean_EMA"):
                buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
            elif name.endswith("variance_EMA"):
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
        for name, p in m.named_parameters():
            if name.endswith("gamma"):
                p.copy_(0.75 + 0.5 * torch.rand(p.shape, generator=g))
    return m.to(dev).eval().prepare()


def _pairs(seed=77, n=N, pairs=PAIRS, m=M, extent=12.0):
    """Registered cloud pairs: positive = anchor @ R + jitter in the SAME point order; half of the positive's keypoints
    are the anchor's (true correspondences), half are drawn independently (negatives within the search radius)."""
    rng = np.random.default_rng(seed)
    anc = (rng.random((pairs, n, 3), dtype=np.float32) * extent).astype(np.float32)
    Rm = np.zeros((pairs, 3, 3), np.float32)
    for b in range(pairs):
        a = rng.uniform(0, 2 * np.pi)
        Rm[b] = np.array([[np.cos(a), np.sin(a), 0], [-np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    pos = (np.matmul(anc, Rm) + rng.normal(0, 0.02, anc.shape)).astype(np.float32)
    ia = np.stack([rng.permutation(n)[:m] for _ in range(pairs)]).astype(np.int32)
    ip = ia.copy()
    ip[:, m // 2:] = np.stack([rng.permutation(n)[: m - m // 2] for _ in range(pairs)])
    return np.concatenate([anc, pos]), Rm, np.concatenate([ia, ip])


def _T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("preset", ["basic_config", "detection_config"])
def test_local_training_forward_vs_oracle(dev, preset):
    from dh3d_amd.training import LocalTrainer
    from oracle import model_np
    m = _build(dev, preset)
    pts, Rm, idx = _pairs()
    w0 = _weights_np(m)
    exp_loss, exp, upd = model_np.local_training_step_forward(pts, Rm, idx, w0, dict(m.config))
    tr = LocalTrainer(m, graph_step=False)
    tr.keep_grads = True
    loss = tr.forward_loss(_T(pts, dev), _T(Rm, dev), _T(idx, dev))
    outs = tr.last_outs
    torch.cuda.synchronize()
    feat = outs["feat"].detach().cpu().numpy()
    scale = float(np.abs(exp["feat"]).max())
    assert np.abs(feat - exp["feat"]).max() <= 1e-4 * scale + 1e-5, np.abs(feat - exp["feat"]).max() / scale
    assert np.abs(outs["local_desc"].detach().cpu().numpy() - exp["local_desc"]).max() <= 1e-4
    assert np.array_equal(outs["xyz_sampled"].cpu().numpy(), exp["xyz_sampled"])
    assert np.abs(outs["feat_sampled"].detach().cpu().numpy() - exp["feat_sampled"]).max() <= 1e-4
    if m.config.detection:
        assert np.abs(outs["attention"].detach().cpu().numpy() - exp["attention"]).max() <= 1e-4
    assert abs(float(loss) - exp_loss) <= 1e-4 * max(1.0, abs(exp_loss)), (float(loss), exp_loss)
    assert exp_loss > 0.05  # positives and negatives both present: a loss that means something
    from dh3d_amd.model import tf_variable_name
    sd = {tf_variable_name(k): v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    assert len(upd) == (22 if m.config.detection else 16), sorted(upd)
    for name, e in upd.items():
        assert not np.array_equal(sd[name], w0[name]), name  # it moved
        assert np.allclose(sd[name], e, rtol=1e-4, atol=1e-6), (name, np.abs(sd[name] - e).max())


# ---------------------------------------------------------------------------------------- float64 torch restatement
def _gather(x, nbr):  # x [B,N,C], nbr [B,N,K] -> [B,N,K,C]
    B, Nn, K = nbr.shape
    return torch.gather(x.unsqueeze(1).expand(B, Nn, x.shape[1], x.shape[2]), 2,
                        nbr.long().unsqueeze(-1).expand(B, Nn, K, x.shape[2]))


def _bn_t(x, bn, eps):  # batch statistics over all rows, biased variance (training mode)
    mu = x.mean((0, 1), keepdim=True)
    var = ((x - mu) ** 2).mean((0, 1), keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * bn.gamma.double() + bn.beta.double()


def _flex_t(x, xyz, nbr, fc):
    fn = _gather(x, nbr)                                                         # [B,N,K,Din]
    dp = _gather(xyz, nbr) - xyz.unsqueeze(2)                                    # [B,N,K,3]
    out = fn.sum(2) @ fc.position_bias.double()
    for d in range(3):
        out = out + (dp[..., d:d + 1] * fn).sum(2) @ fc.position_theta[d].double()
    return out + fc.feature_bias.double().reshape(1, 1, -1)


def _stack_t(mod, x, xyz, nbr):
    for i in range(len(mod.outdims)):
        fc, bn = getattr(mod, "flexconv_%d" % i), getattr(mod, "flexconv_%d_bn" % i)
        x = torch.relu(_bn_t(_flex_t(x, xyz, nbr, fc), bn, bn.eps))
    pool = _gather(x, nbr).max(2).values
    f1, f2 = mod.se.f1.tfconv0, mod.se.f2.tfconv0
    sq = torch.relu(pool @ f1.W.double().reshape(f1.cin, f1.cout) + f1.b.double())
    g = torch.sigmoid(sq @ f2.W.double().reshape(f2.cin, f2.cout) + f2.b.double())
    return torch.relu(x + x * g)


def _conv_t(x, fc1d):
    c = fc1d.tfconv0
    return torch.relu(_bn_t(x @ c.W.double().reshape(c.cin, c.cout) + c.b.double(), c.bn, c.bn.eps))


def _restated_loss(m, pts, R, idx, lv, nbr):
    """The same graph in float64 tensor ops (integer geometry taken from the HIP forward: it is exact)."""
    from dh3d_amd import losses
    xyz = pts.double()
    dp = _gather(xyz, nbr) - _gather(xyz, nbr)[:, :, 0:1]
    ic = m.initconv
    init = dp.sum(2) @ ic.position_theta.double() + ic.position_bias.double()
    init = torch.relu(_bn_t(init, m.initconv_bn, m.initconv_bn.eps))
    init = _gather(init, nbr).max(2).values
    x1 = _stack_t(m.stage1, init, xyz, nbr)
    x2 = _conv_t(x1, m.before_stage2_conv1d)
    idxs = lv["idx"].long()
    fs = torch.gather(x2, 1, idxs.unsqueeze(-1).expand(-1, -1, x2.shape[2]))
    y = _stack_t(m.stage2, fs, lv["xyz_s"].double(), lv["nbr_s"])
    d = torch.clamp(lv["nn3_dist"].double(), min=1e-10)
    wts = (1.0 / d) / (1.0 / d).sum(2, keepdim=True)
    up = (_gather(y, lv["nn3_idx"]) * wts.unsqueeze(-1)).sum(2)
    x2 = _conv_t(torch.cat([up, x2], 2), m.stage2.concat_conv1d)
    feat = _conv_t(x1, m.local_stage1_shortcut) + x2
    desc = feat * torch.rsqrt(torch.clamp((feat * feat).sum(2, keepdim=True), min=1e-8))
    kp = idx.long()
    take = lambda t: torch.gather(t, 1, kp.unsqueeze(-1).expand(-1, -1, t.shape[2]))
    # (coordinates stay float32: the losses' masks -- which pairs count as positives / negatives -- are then the very
    #  same booleans the HIP step computed)
    outs = {"xyz": pts, "feat": feat, "local_desc": desc, "R": R, "sample_nodes_concat": idx.reshape(idx.shape[0], -1, 1),
            "xyz_sampled": take(pts), "feat_sampled": take(desc)}
    if m.config.detection:
        det = m.detection_block_reliable
        x = feat
        for i in range(len(det.conv_dims)):
            c = getattr(det, "detec_conv%d" % i)
            x = torch.relu(_bn_t(x @ c.W.double().reshape(c.cin, c.cout) + c.b.double(), c.bn, c.bn.eps))
        fcw = det.detec_conv_fc
        att = torch.sigmoid(x @ fcw.W.double().reshape(-1, 1) + fcw.b.double())
        outs["attention"], outs["att_sampled"] = att, take(att)
    return losses.compute_loss(outs, m.config)


@pytest.mark.parametrize("preset", ["basic_config", "detection_config"])
def test_local_training_gradients_vs_float64_restatement(dev, preset):
    from dh3d_amd.training import LocalTrainer, local_trainable_parameters
    pts, Rm, idx = _pairs(seed=78)
    tp, tR, ti = _T(pts, dev), _T(Rm, dev), _T(idx, dev)
    m = _build(dev, preset, seed=6)
    names = {id(p): n for n, p in m.named_parameters()}
    tr = LocalTrainer(m, graph_step=False, weight_decay=0.0)
    loss = tr.forward_loss(tp, tR, ti)
    loss.backward()
    params = local_trainable_parameters(m)
    got = [(names[id(p)], p.grad.detach().double().clone()) for p in params if p.grad is not None]
    for p in params:
        p.grad = None
    with torch.no_grad():
        geo = m._geometry(tp, None)
        m._join_side(geo)
        lv = geo.level(8, 8)
        nbr = geo.nbr
    ref = _restated_loss(m, tp, tR, ti, lv, nbr)
    assert abs(float(ref) - float(loss)) <= 1e-4 * max(1.0, abs(float(ref))), (float(ref), float(loss))
    ref.backward()
    exp = {names[id(p)]: p.grad.detach().double().clone() for p in params if p.grad is not None}
    assert len(got) == len(exp) >= (34 if preset == "basic_config" else 44), (len(got), len(exp))
    top = max(float(v.abs().max()) for v in exp.values())
    report = []
    for n, a in got:
        b = exp[n]
        # (biases in front of a BatchNorm have an exactly-zero gradient: the floor keeps them from being compared
        #  relative to their own rounding noise)
        scale = max(float(b.abs().max()), 1e-4 * top)
        err = float((a - b).abs().max()) / scale
        report.append((err, n, scale / top))
    print("local training gradient errors (relative to the tensor's largest entry; tensor scale / largest gradient):",
          [(round(e, 5), n, round(r, 5)) for e, n, r in sorted(report, reverse=True)[:8]])
    for err, n, _ in report:
        assert err <= TOL_GRAD, (n, err)


def test_local_trainer_whole_step_graph_follows_eager_steps_and_learns(dev):
    from dh3d_amd.training import LocalTrainer
    batches = [tuple(_T(a, dev) for a in _pairs(seed=s, n=2048, m=128)) for s in (90, 91)]
    traj = []
    for graph in (True, False):
        m = _build(dev, "detection_config", seed=9)
        tr = LocalTrainer(m, start_lr=2e-4, graph_step=graph)
        ls = [tr.step(*batches[i % 2]) for i in range(8)]
        assert bool(tr._graphs) == graph
        traj.append(ls)
    assert all(np.isfinite(traj[0])) and all(np.isfinite(traj[1]))
    for x, y in zip(*traj):
        assert abs(x - y) <= 3e-2 * max(1.0, abs(y)), traj
    # a longer run on one batch: the descriptors of corresponding keypoints move together
    m = _build(dev, "basic_config", seed=10)
    tr = LocalTrainer(m, start_lr=1e-3)
    ls = [tr.step(*batches[0]) for _ in range(40)]
    assert np.mean(ls[-5:]) < 0.8 * np.mean(ls[:5]), (ls[:5], ls[-5:])
    # the inference path sees the trained weights and moving averages
    m.eval()
    with torch.no_grad():
        o = m(batches[0][0], fetch=("xyz_feat",))
    assert torch.isfinite(o["xyz_feat"]).all()

"""Losses against the numpy restatement oracle/losses_np.py of core/losses.py.

CPU: the tensor math on CPU tensors (host logic).  GPU (-m gpu): the same on device tensors -- what the training
step evaluates -- incl. the detector loss whose 16-NN runs on the kNN kernel, and the config-driven assembly."""
import numpy as np
import pytest
import torch


def _descs(rng, B, P, Ng, other, D=256):
    n = B * (1 + P + Ng + (1 if other else 0))
    d = rng.standard_normal((n, D)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # make some triplets violated and some satisfied: pull the positives towards their anchor
    d[B:B + P * B] = 0.7 * d[B:B + P * B] + 0.3 * np.repeat(d[:B], P, axis=0)
    return d


def _pair_outs(rng, B=2, N=600, M=64, D=32):
    """A registered cloud pair batch [cloud0 x B | cloud1 x B] with sampled keypoints (losses.py:29-36,66-75)."""
    xyz0 = (rng.random((B, N, 3)) * 20 - 10).astype(np.float32)
    th = rng.random(B) * 6.28
    R = np.stack([np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]]) for t in th]).astype(np.float32)
    xyz1 = (np.matmul(xyz0, R) + 0.05 * rng.standard_normal((B, N, 3))).astype(np.float32)
    perm = np.stack([rng.permutation(N) for _ in range(B)])
    xyz1 = np.take_along_axis(xyz1, perm[:, :, None], 1)
    feat0 = rng.standard_normal((B, N, D)).astype(np.float32)
    feat1 = np.take_along_axis(feat0 + 0.3 * rng.standard_normal((B, N, D)).astype(np.float32), perm[:, :, None], 1)
    feat0 /= np.linalg.norm(feat0, axis=2, keepdims=True); feat1 /= np.linalg.norm(feat1, axis=2, keepdims=True)
    s0 = np.stack([rng.choice(N, M, replace=False) for _ in range(B)])
    inv = np.argsort(perm, axis=1)
    s1 = np.take_along_axis(inv, s0, 1)  # the matching points in cloud 1
    s1[:, ::3] = rng.integers(0, N, (B, (M + 2) // 3))  # a third of them mismatched
    xyz = np.concatenate([xyz0, xyz1]); feat = np.concatenate([feat0, feat1])
    samp = np.concatenate([s0, s1])[:, :, None].astype(np.int32)
    b = np.arange(2 * B)[:, None]
    return {"xyz": xyz, "feat": feat, "sample_nodes_concat": samp, "R": R,
            "xyz_sampled": xyz[b, samp[:, :, 0]], "feat_sampled": feat[b, samp[:, :, 0]],
            "att_sampled": rng.random((2 * B, M, 1)).astype(np.float32)}


def _to(outs, dev):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in outs.items()}


def _check_global_and_local(dev):
    from dh3d_amd import losses
    from oracle import losses_np
    rng = np.random.default_rng(13)
    for (B, P, Ng) in [(1, 2, 18), (2, 2, 8), (3, 1, 4)]:
        d = _descs(rng, B, P, Ng, True)
        t = torch.from_numpy(d).to(dev)
        exp = losses_np.lazy_quadruplet_loss(d, B, P, Ng, 0.5, 0.2)
        got = float(losses.lazy_quadruplet_loss(t, B, P, Ng, 0.5, 0.2))
        assert abs(got - exp) < 1e-5, (B, P, Ng, got, exp)
        d3 = d[:B * (1 + P + Ng)]
        exp = losses_np.lazy_triplet_loss(d3, B, P, Ng, 0.5)
        got = float(losses.lazy_triplet_loss(torch.from_numpy(d3).to(dev), B, P, Ng, 0.5))
        assert abs(got - exp) < 1e-5
    outs = _pair_outs(rng)
    for kw in (dict(), dict(pos_r=0.5, search_r=20.0, margin=1.0, neg_weight=5.0)):  # defaults / basic_config values
        exp = losses_np.desc_local_loss(outs, **kw)
        got = float(losses.desc_local_loss(_to(outs, dev), **kw))
        assert abs(got - exp) <= 1e-5 * max(1.0, abs(exp)), (kw, got, exp)


def test_global_and_local_losses_cpu():
    _check_global_and_local(torch.device("cpu"))


def test_compute_loss_is_config_driven_like_the_reference():
    """core/model.py:212-237: loss(outs, **config) with the config's own margins / radii and *_loss_weight."""
    from dh3d_amd import ConfigFactory, losses
    from oracle import losses_np
    rng = np.random.default_rng(5)
    cfg = ConfigFactory("global_config").getconfig()
    assert cfg.margin == 1.0 and cfg.neg_weight == 5.0 and cfg.pos_r == 0.5 and cfg.search_r == 20.0
    det = ConfigFactory("detection_config").getconfig()
    assert det.ar_th == 0.4 and det.det_k == 16 and det.det_loss_weight == 0.2 and det.pos_r == 0.5
    B, P, Ng = cfg.batch_size, cfg.num_pos, cfg.num_neg
    d = _descs(rng, B, P, Ng, True)
    cfg.global_loss_weight = 0.5
    got = float(losses.compute_loss({"global_desc": torch.from_numpy(d)}, cfg))
    assert abs(got - 0.5 * losses_np.lazy_quadruplet_loss(d, **cfg)) < 1e-5
    loc = ConfigFactory("basic_config").getconfig()
    outs = _pair_outs(rng)
    got = float(losses.compute_loss(_to(outs, "cpu"), loc))
    exp = losses_np.desc_local_loss(outs, **loc)
    assert abs(got - exp) <= 1e-5 * max(1.0, abs(exp))


@pytest.mark.gpu
def test_losses_on_device_vs_oracle(dev):
    from dh3d_amd import ConfigFactory, losses
    from oracle import losses_np
    _check_global_and_local(dev)
    rng = np.random.default_rng(29)
    outs = _pair_outs(rng, B=2, N=700, M=96)
    det = ConfigFactory("detection_config").getconfig()
    for kw in (dict(), dict(ar_th=det.ar_th, det_k=det.det_k, ar_nn_k=det.ar_nn_k, pos_r=det.pos_r),
               dict(use_hardest_neg=False)):
        exp = losses_np.local_detection_loss_nn(outs, **kw)
        got = float(losses.local_detection_loss_nn(_to(outs, dev), **kw))
        assert abs(got - exp) < 1e-5, (kw, got, exp)
    exp = det.local_loss_weight * losses_np.desc_local_loss(outs, **det) + \
        det.det_loss_weight * losses_np.local_detection_loss_nn(outs, **det)
    got = float(losses.compute_loss(_to(outs, dev), det))
    assert abs(got - exp) < 1e-5 * max(1.0, abs(exp))


@pytest.mark.gpu
def test_pairwise_sqdist_kernel_matches_the_broadcast_form(dev):
    """train_ops.pairwise_sqdist (HIP forward, closed-form backward on the batched GEMM kernels) == the broadcast
    restatement of core/tf_utils.py:125-136 in float64, values and both gradients; ragged sizes, near-identical rows
    (the local losses take sqrt(d + 1e-10) of such entries: no |a|^2 + |b|^2 - 2ab cancellation is allowed)."""
    import torch
    from dh3d_amd import train_ops as T
    g = torch.Generator().manual_seed(12)
    for (B, n, m, D) in ((3, 70, 45, 128), (2, 512, 512, 128), (1, 33, 31, 16)):
        a = torch.randn(B, n, D, generator=g)
        a = a / a.norm(dim=2, keepdim=True)
        b = torch.randn(B, m, D, generator=g)
        b = b / b.norm(dim=2, keepdim=True)
        k = min(n, m) // 2
        b[:, :k] = a[:, :k] + 1e-4 * torch.randn(B, k, D, generator=g)   # near-identical pairs
        ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        out = T.pairwise_sqdist(ad, bd)
        a64, b64 = a.double().to(dev).requires_grad_(True), b.double().to(dev).requires_grad_(True)
        ref = ((a64.unsqueeze(2) - b64.unsqueeze(1)) ** 2).sum(3)
        err = (out.double() - ref).abs()
        assert float((err / (ref + 1e-9)).max()) < 1e-4 and float(err.max()) < 1e-5, (float(err.max()), B, n, m, D)
        w = torch.randn(B, n, m, generator=g).to(dev)
        (torch.sqrt(out + 1e-10) * w).sum().backward()
        (torch.sqrt(ref + 1e-10) * w.double()).sum().backward()
        for x, y in ((ad.grad, a64.grad), (bd.grad, b64.grad)):
            assert float((x.double() - y).abs().max()) <= 2e-3 * float(y.abs().max()), (B, n, m, D)

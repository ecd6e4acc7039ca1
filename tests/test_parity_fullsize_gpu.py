"""GPU: END-TO-END parity of the HIP forward against the CPU oracle (oracle/model_np.py + the C ops) at the sizes the
bench is quoted on -- cfg2 (local, N=8192), cfg3 (global, N=4096, several clouds), cfg5 (N=16384, with and without
host kNN indices) -- and at the cloud sizes where the kernel dispatch changes (4096 / 12288 / 16384 boundaries).

Bars (BASELINE.json north_star / SURVEY 8c):
  * kNN ids, FPS picks, sampled-set kNN ids, three_nn ids: bit-equal;
  * un-normalised features: |a-b| <= atol + 1e-4*|b| (the reference's own assertAllClose(…, 1e-4) form,
    user_ops/misc.py:89-97) with atol = 1e-5 x the tensor's largest magnitude: the reference's atol of 1e-6 is for
    O(1) single-op outputs, these are O(10) sums after ten stacked layers;
  * L2-normalised descriptors (local 'xyz_feat', global 'globaldesc'): within 1e-4 absolute.
BatchNorm statistics are randomised so that a folded-BN slip cannot hide behind identity statistics.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _weights_np(model):
    from dh3d_amd.model import tf_variable_name
    return {tf_variable_name(k): v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def _randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, buf in model.named_buffers():
            if name.endswith(("mean_EMA", "moving_mean")):
                buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
            elif name.endswith(("variance_EMA", "moving_variance")):
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
        for name, p in model.named_parameters():
            if name.endswith("gamma"):
                p.copy_(0.75 + 0.5 * torch.rand(p.shape, generator=g))


def _build(preset, dev, seed=0, num_points=None):
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    cfg = ConfigFactory(preset).getconfig()
    if num_points is not None:
        cfg.num_points = num_points
    m = DH3D(cfg).init_synthetic(seed)
    _randomise_bn(m, seed + 1)
    return m.to(dev).eval().prepare()


def _feat_close(got, exp):
    """the reference's CPU-vs-GPU criterion, atol scaled to the tensor (features are O(1..10), not normalised)"""
    scale = float(np.abs(exp).max())
    err = np.abs(got - exp)
    ok = err <= 1e-5 * scale + 1e-4 * np.abs(exp)
    return bool(ok.all()), float(err.max() / max(scale, 1e-30))


def _check_level(model, trace, scope):
    """FPS picks, sampled-set kNN ids, three_nn ids of the shared N/8 level: bit-equal."""
    lv = model._last_geo._lv
    assert np.array_equal(lv["idx"].cpu().numpy(), trace[scope + "/fps_idx"]), "FPS picks differ"
    assert np.array_equal(lv["nbr_s"].cpu().numpy(), trace[scope + "/knn"].transpose(0, 2, 1)), "N/8 kNN ids differ"
    assert np.array_equal(lv["nn3_idx"].cpu().numpy(), trace[scope + "/nn3_idx"]), "three_nn ids differ"
    assert np.array_equal(lv["nn3_dist"].cpu().numpy(), trace[scope + "/nn3_dist"]), "three_nn distances differ"


def _run_and_compare(m, pts, dev, knn_inds=None):
    from oracle import model_np
    trace = {}
    exp = model_np.forward(pts, _weights_np(m), detection=bool(m.config.detection),
                           extract_global=bool(m.config.extract_global),
                           knn_inds=None if knn_inds is None else knn_inds.cpu().numpy(), trace=trace)
    tp = torch.from_numpy(pts).to(dev)
    with torch.no_grad():
        outs = m(tp, knn_inds=knn_inds)
    torch.cuda.synchronize()
    assert np.array_equal(outs["knn_inds"].cpu().numpy(), exp["knn_indices"].transpose(0, 2, 1)), "kNN ids differ"
    _check_level(m, trace, "stage2")
    ok, rel = _feat_close(outs["feat"].cpu().numpy(), exp["feat"])
    assert ok, ("feat", rel)
    xf = outs["xyz_feat"].cpu().numpy()
    assert np.array_equal(xf[:, :, :3], pts)
    err = float(np.abs(xf - exp["xyz_feat"]).max())
    assert err < 1e-4, ("xyz_feat", err)
    if m.config.detection:
        e = float(np.abs(outs["xyz_feat_att"].cpu().numpy() - exp["xyz_feat_att"]).max())
        assert e < 1e-4, ("xyz_feat_att", e)
    if m.config.extract_global:
        _check_level(m, trace, "global_before_assemble")
        g = outs["globaldesc"].cpu().numpy()
        e = float(np.abs(g - exp["globaldesc"]).max())
        assert e < 1e-4, ("globaldesc", e)
    return outs, exp


def test_cfg2_local_forward_N8192_vs_oracle(dev):
    """BASELINE config[1]: basic_config, N=8192, K=8 -- every >=4096-point kernel of the bench path (Morton kNN,
    batched FPS, flex_conv x6, linear x6, fused up-sample + concat conv) against the oracle; both the all-outputs
    forward and the fetch=('xyz_feat',) forward the bench replays (fused l2-normalise store)."""
    m = _build("basic_config", dev, seed=21)
    pts = np.random.default_rng(2002).random((1, 8192, 3), dtype=np.float32)
    outs, exp = _run_and_compare(m, pts, dev)
    with torch.no_grad():
        only = m(torch.from_numpy(pts).to(dev), fetch=("xyz_feat",))
        run = m.graphed(torch.from_numpy(pts).to(dev), outputs=("xyz_feat",))
        rep = run()["xyz_feat"].clone()
    for name, t in (("fetch", only["xyz_feat"]), ("graph replay", rep)):
        e = float(np.abs(t.cpu().numpy() - exp["xyz_feat"]).max())
        assert e < 1e-4, (name, e)


def test_cfg2_detection_head_N4096_vs_oracle(dev):
    """detection_config at a size where the detector's wide layer runs on the tiled bf16x6 head."""
    m = _build("detection_config", dev, seed=22)
    pts = np.random.default_rng(2003).random((1, 4096, 3), dtype=np.float32)
    _run_and_compare(m, pts, dev)


def test_cfg3_global_forward_N4096_vs_oracle(dev):
    """BASELINE config[2]: global_config, N=4096, several clouds (uniform cube and an Oxford-like +-20 m extent):
    shortcut fused into the concat conv, commuted attention head, NetVLAD."""
    m = _build("global_config", dev, seed=23)
    rng = np.random.default_rng(3003)
    pts = rng.random((3, 4096, 3), dtype=np.float32)
    pts[2] = pts[2] * 40 - 20
    outs, exp = _run_and_compare(m, pts, dev)
    with torch.no_grad():
        run = m.graphed(torch.from_numpy(pts).to(dev), outputs=("globaldesc",))
        rep = run()["globaldesc"].clone()
    e = float(np.abs(rep.cpu().numpy() - exp["globaldesc"]).max())
    assert e < 1e-4, ("graph replay globaldesc", e)


def _scene(B, N, seed):
    """A street scene normalised to [-1, 1] (ground plane + six walls + clutter, z extent a tenth of x / y): the shape of the
    reference's data, where the sort flags the cloud as crowded (kNN on the pruned scan), FPS picks meet near-ties on the
    plane and the Morton boxes are flat -- every other end-to-end test runs on the uniform cube."""
    rng = np.random.default_rng(seed)
    out = np.empty((B, N, 3), np.float32)
    for b in range(B):
        n_g, n_w = int(N * 0.55), int(N * 0.35)
        g = np.stack([rng.uniform(-1, 1, n_g), rng.uniform(-1, 1, n_g), rng.normal(-0.08, 0.004, n_g)], 1)
        walls = []
        for _ in range(6):
            m = n_w // 6
            x0, y0, ang, ln = rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0, np.pi), rng.uniform(0.3, 0.9)
            t = rng.uniform(0, ln, m)
            walls.append(np.stack([x0 + t * np.cos(ang), y0 + t * np.sin(ang), rng.uniform(-0.08, 0.12, m)], 1)
                         + rng.normal(0, 0.003, (m, 3)))
        w = np.concatenate(walls)
        c = rng.uniform(-1, 1, (N - n_g - len(w), 3)) * np.array([1, 1, 0.1])
        out[b] = np.clip(np.concatenate([g, w, c])[rng.permutation(N)], -1, 1)
    return out


def test_scene_like_clouds_local_N8192_and_global_N4096_vs_oracle(dev):
    """The same end-to-end bars on scene-like clouds: kNN / FPS / sampled-set kNN / three_nn ids bit-equal, descriptors
    within 1e-4 -- the local forward at N = 8192 and the global forward at N = 4096 (one scene and one uniform cloud in a
    batch: the sort's per-cloud verdict sends one to the pruned scan and the other to the cell lists)."""
    from dh3d_amd import pm
    m = _build("basic_config", dev, seed=31)
    pts = _scene(1, 8192, 77)
    _run_and_compare(m, pts, dev)
    assert int(pm.spatial_sort_cells(torch.from_numpy(pts).to(dev))[2][0, 4106]) == 1  # (this IS a crowded cloud)
    m = _build("global_config", dev, seed=32)
    pts = np.concatenate([_scene(1, 4096, 78), np.random.default_rng(79).random((1, 4096, 3), dtype=np.float32)])
    flags = pm.spatial_sort_cells(torch.from_numpy(pts).to(dev))[2][:, 4106].cpu().tolist()
    assert flags == [1, 0]
    _run_and_compare(m, pts, dev)


def test_cfg5_save_all_forward_N16384_vs_oracle(dev):
    """BASELINE config[4]: N=16384 (save_all path, localdesc_extract.py:146,166): with host-style kNN indices as the
    reference requires above 8192 points (core/model.py:148-155) and with the device kNN (superset)."""
    from dh3d_amd import pm
    m = _build("basic_config", dev, seed=24, num_points=16384)
    pts = np.random.default_rng(5005).random((1, 16384, 3), dtype=np.float32)
    outs, exp = _run_and_compare(m, pts, dev)           # device kNN
    nbr = outs["knn_inds"].clone()
    outs2, _ = _run_and_compare(m, pts, dev, knn_inds=nbr)  # indices as an input
    assert torch.equal(outs2["xyz_feat"], outs["xyz_feat"])


@pytest.mark.parametrize("N", [4095, 4096, 4097, 12288, 12289])
def test_dispatch_thresholds_vs_oracle(dev, N):
    """Cloud sizes either side of the rules that pick kernels (batched FPS 4096..12288, linear_x6 / fused stores /
    commuted head from 4096 points per cloud): every side must agree with the oracle, ids bit-equal."""
    m = _build("global_config", dev, seed=30 + N % 7)
    pts = np.random.default_rng(N).random((1, N, 3), dtype=np.float32)
    _run_and_compare(m, pts, dev)


def test_netvlad_cfg3_shape_vs_restatement(dev):
    """NetVLAD + context gating at cfg3's (B=32, N=4096): the chunking netvlad_chunks() picks there (8 chunks of 8
    tiles) is not reached by the small cases of test_pm_gpu.py."""
    from oracle import model_np
    from dh3d_amd import pm
    B, N, D, C, O = 32, 4096, 256, 64, 256
    rng = np.random.default_rng(324096)
    x = rng.standard_normal((B, N, D)).astype(np.float32)
    att = rng.random((B, N, 1), dtype=np.float32)
    w = {"cluster_weights": (rng.standard_normal((D, C)) / 16).astype(np.float32),
         "cluster_weights2": (rng.standard_normal((1, D, C)) / 16).astype(np.float32),
         "hidden1_weights": (rng.standard_normal((D * C, O)) / 8).astype(np.float32),
         "gating_weights": (rng.standard_normal((O, O)) / 16).astype(np.float32)}
    for s, n in (("cluster_bn", C), ("bn", O), ("gating_bn", O)):
        w[s + "/gamma"] = (0.5 + rng.random(n)).astype(np.float32)
        w[s + "/beta"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        w[s + "/moving_mean"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        w[s + "/moving_variance"] = (0.5 + rng.random(n)).astype(np.float32)
    exp = model_np.global_netvlad_block(x, att, w, 1e-3)

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def fold(s):
        sc = w[s + "/gamma"] / np.sqrt(w[s + "/moving_variance"] + np.float32(1e-3))
        return T(sc), T(w[s + "/beta"] - w[s + "/moving_mean"] * sc)
    cs, ch = fold("cluster_bn"); s1, h1 = fold("bn"); s2, h2 = fold("gating_bn")
    vlad = pm.netvlad_aggregate(T(x), T(att), pm.pack_weight(T(w["cluster_weights"])), cs, ch,
                                T(w["cluster_weights2"].reshape(D, C)))
    out = pm.netvlad_head(vlad, T(w["hidden1_weights"]), s1, h1, T(w["gating_weights"]), s2, h2).cpu().numpy()
    scale = np.abs(exp).max()
    assert np.abs(out - exp).max() <= 1e-4 * scale, np.abs(out - exp).max() / scale
    outn = pm.netvlad_head(vlad, T(w["hidden1_weights"]), s1, h1, T(w["gating_weights"]), s2, h2,
                           l2_eps=1e-8).cpu().numpy()
    assert np.abs(outn - exp / np.linalg.norm(exp, axis=1, keepdims=True)).max() < 1e-4


def test_forward_above_16384_points_with_host_knn_vs_oracle(dev):
    """num_points > 16384: the reference takes any such size with host kNN indices (core/model.py:38,148-155); here the
    FPS then keeps its running distances in scratch (any-N kernel).  N = 18000 against the oracle."""
    m = _build("basic_config", dev, seed=41, num_points=18000)
    pts = np.random.default_rng(18000).random((1, 18000, 3), dtype=np.float32)
    from oracle import cpu as O
    nn, _ = O.knn_bruteforce(np.ascontiguousarray(pts.transpose(0, 2, 1)), 8)
    nbr = torch.from_numpy(nn).to(dev)
    _run_and_compare(m, pts, dev, knn_inds=nbr)
    bad = nbr.clone(); bad[0, 5, 3] = 18000
    with pytest.raises(ValueError):
        m(torch.from_numpy(pts).to(dev), knn_inds=bad)       # out-of-range ids are refused, not dereferenced
    with pytest.raises(ValueError):
        m(torch.from_numpy(pts).to(dev), knn_inds=nbr[:, :100])


@pytest.mark.parametrize("N", [20000, 40000])
def test_device_knn_above_16384_points_vs_oracle(dev, N):
    """no host round trip for the neighbours of ANY cloud size (the reference: host sklearn indices above
    8192 points, core/utils.py:53-57).  N = 20000: the whole forward with the device kNN against the oracle; N = 40000:
    the kNN ids + distances against the oracle's brute force (1.6 G pairs of scalar C), ties included (a duplicated
    block of points), and the forward on them finite with exact pass-through coordinates."""
    from oracle import cpu as O
    from dh3d_amd import pm
    rng = np.random.default_rng(N)
    pts = rng.random((1, N, 3), dtype=np.float32)
    pts[0, N - 50:] = pts[0, 100:150]  # exact duplicates: the CUB tie rule with the continued ladder (1024, ceil(N/1024))
    m = _build("basic_config", dev, seed=44, num_points=N)
    if N <= 20000:
        _run_and_compare(m, pts, dev)  # knn_inds=None: the device search
        return
    tp = torch.from_numpy(pts).to(dev)
    nbr, dist = pm.knn_xyz(tp, 8)
    nn, dd = O.knn_bruteforce(np.ascontiguousarray(pts.transpose(0, 2, 1)), 8)
    assert np.array_equal(nbr.cpu().numpy(), nn)
    assert np.array_equal(dist.cpu().numpy().view(np.uint32), dd.view(np.uint32))
    with torch.no_grad():
        o = m(tp, fetch=("xyz_feat", "knn_inds"))
    assert torch.equal(o["knn_inds"], nbr) and torch.isfinite(o["xyz_feat"]).all()
    assert torch.equal(o["xyz_feat"][:, :, :3], tp)


def test_fps_contract_switch_reaches_the_model(dev):
    """config.fps_contract = 0: the whole forward on the uncontracted FPS rounding (integrator's switch), against the
    oracle run in the same mode."""
    from oracle import cpu as O
    m = _build("basic_config", dev, seed=43)
    m.config.fps_contract = 0
    pts = (np.random.default_rng(43).random((1, 4096, 3), dtype=np.float32) * 40 - 20)
    with torch.no_grad():
        m(torch.from_numpy(pts).to(dev))
    got = m._last_geo._lv["idx"].cpu().numpy()
    assert np.array_equal(got, O.farthest_point_sample(512, pts, contract=False))

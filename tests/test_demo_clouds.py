"""The reference's OWN inputs on this path -- evaluate/local_eval/demo_data/{268,642}.bin (cfg 5: 16384-point Oxford LiDAR
sub-maps in metres), the oxford_dso cloud padded to 9000 points with duplicates (core/utils.py:103-105), and clouds of
evaluate/global_eval/demo_data cropped the way Global_test_dataset crops them -- frozen as data with the oracle's
outputs in tests/golden/demo_clouds.npz (tests/golden/make_demo_golden.py).  Every other end-to-end test runs on
seeded uniform cubes or the synthetic `_scene()`; these are real street clouds: a ground plane that holds most points,
50 m extent, exact duplicates.

CPU tests pin the oracle on the fixture (and, through the reference's own scipy recipe, the kNN ids of a real cloud);
the GPU tests run the HIP path: ids bit-equal, descriptors within 1e-4, NMS keypoints against the float64 restatement.
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LOCAL = ("local_268", "local_642", "dso_9000")
GLOBAL = ("global_a", "global_b", "global_c")


@pytest.fixture(scope="module")
def demo():
    return np.load(os.path.join(HERE, "golden", "demo_clouds.npz"))


def _weights(preset, demo, seed=0):
    import torch
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D, tf_variable_name
    m = DH3D(ConfigFactory(preset).getconfig()).init_synthetic(seed)
    # the fixture's calibrated BatchNorm statistics (make_demo_golden.py: one training-mode pass on a demo cloud)
    sd = m.state_dict()
    prefix = "bn/%s/" % preset
    cal = {k[len(prefix):]: demo[k] for k in demo.files if k.startswith(prefix)}
    hit = 0
    for k in sd:
        if tf_variable_name(k) in cal:
            sd[k] = torch.from_numpy(cal[tf_variable_name(k)].reshape(sd[k].shape))
            hit += 1
    assert hit == len(cal) > 10
    m.load_state_dict(sd)
    w = {tf_variable_name(k): v.detach().numpy() for k, v in m.state_dict().items()}
    chk = sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values())
    assert abs(chk - float(demo["weights_checksum_" + preset])) <= 1e-9 * chk, "init_synthetic(0) is not the fixture's"
    return m, w


# ------------------------------------------------------------------------------------------------ CPU: the oracle, pinned
def test_fixture_is_the_reference_data(demo):
    for name, n in (("local_268", 16384), ("local_642", 16384), ("dso_9000", 9000), ("global_a", 4096),
                    ("global_b", 4096), ("global_c", 8192)):
        c = demo[name]
        assert c.shape == (n, 3) and c.dtype == np.float32 and np.isfinite(c).all()
        assert np.abs(c).max() > 8.0           # metres, not a unit cube
    assert len(np.unique(demo["dso_9000"], axis=0)) == 8920      # 80 duplicates by padding
    assert len(np.unique(demo["local_268"], axis=0)) == 16384


@pytest.mark.parametrize("name", ["global_a", "dso_9000"])
def test_oracle_ids_on_real_clouds_match_fixture_and_scipy(oracle, demo, name):
    """The oracle today == the frozen ids; and the reference's own checker (float64 pdist -> argsort,
    user_ops/test_knn_bruteforce.py:32-40) agrees with them wherever its K-th / K+1-th gap is not a float32 near-tie
    (duplicates make exact ties: there the CUB order = smaller index first is what the oracle states)."""
    from scipy.spatial import cKDTree
    c = demo[name]
    N = len(c)
    nn, dist = oracle.knn_bruteforce(np.ascontiguousarray(c.T[None]), 8)
    assert np.array_equal(nn[0], demo[name + "/knn"])
    d64, i64 = cKDTree(c.astype(np.float64)).query(c.astype(np.float64), k=9)
    gap = d64[:, 1:] - d64[:, :-1]                     # consecutive gaps among the first nine
    scale = np.maximum(d64[:, 1:], 1e-30)
    clear = (gap > 4 * np.finfo(np.float32).eps * np.maximum(scale, np.abs(c).max())).all(1)
    assert clear.mean() > (0.9 if name == "dso_9000" else 0.99)    # (the DSO cloud is quantised: many equal distances)
    assert np.array_equal(nn[0][clear], i64[clear, :8])
    # FPS picks and three_nn of the frozen level
    picks = oracle.farthest_point_sample(N // 8, c[None])
    assert np.array_equal(picks[0], demo[name + "/fps_idx"])
    sub = c[picks[0]]
    d3, i3 = oracle.three_nn(c[None], sub[None])
    assert np.array_equal(i3[0], demo[name + "/nn3_idx"]) and np.array_equal(d3[0], demo[name + "/nn3_dist"])


def test_duplicates_tie_rule_on_the_padded_cloud(demo):
    """A padded duplicate and its original are at distance 0 from each other, so the kNN rows of both start with the
    same pair -- in the order of cub::BlockRadixSort over the blocked arrangement (knn_bruteforce_kernel_gpu.cu.cc:98-123):
    a stable sort, so equal keys keep their linear rank (x % C_THREADS) * C_VPT + x / C_THREADS, with the ladder's
    (1024, 9) at N = 9000 (:181-221).  Rank 0 is therefore NOT always the point itself, nor the smaller index."""
    c, nn = demo["dso_9000"], demo["dso_9000/knn"]
    _, first, inv = np.unique(c, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    dup = np.nonzero(first[inv] != np.arange(len(c)))[0]   # padded copies (index >= 8920)
    assert len(dup) >= 70 and dup.min() >= 8920
    rank = lambda x: (x % 1024) * 9 + x // 1024
    not_self_first = 0
    for j in dup:
        group = sorted(np.nonzero(inv == inv[j])[0], key=rank)
        for member in group:
            assert list(nn[member, : len(group)]) == list(group)
            not_self_first += int(nn[member, 0] != member)
    assert not_self_first >= len(dup)


# ------------------------------------------------------------------------------------------------ GPU: the HIP path
def _forward(model, pts, dev, **kw):
    import torch
    with torch.no_grad():
        outs = model(torch.from_numpy(np.ascontiguousarray(pts)).to(dev), **kw)
    torch.cuda.synchronize()
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("name", LOCAL)
def test_local_demo_cloud_cfg5_path(dev, demo, name):
    """localdesc_extract.py:141-170 on the shipped clouds: detection_config at num_points = 16384 / 9000, device kNN
    (bit-equal to the oracle's ids, i.e. to what the host ball tree + the functor order give) and the host-indices
    input the reference uses above 8192 points; 'xyz_feat_att' rows within 1e-4."""
    import torch
    c = demo[name]
    N = len(c)
    m, _ = _weights("detection_config", demo)
    m.config.num_points = N
    m = m.to(dev).eval().prepare()
    outs = _forward(m, c[None], dev)
    assert np.array_equal(outs["knn_inds"][0].cpu().numpy(), demo[name + "/knn"]), "kNN ids differ"
    lv = m._last_geo._lv
    assert np.array_equal(lv["idx"][0].cpu().numpy(), demo[name + "/fps_idx"]), "FPS picks differ"
    assert np.array_equal(lv["nbr_s"][0].cpu().numpy(), demo[name + "/knn_s"]), "N/8 kNN ids differ"
    assert np.array_equal(lv["nn3_idx"][0].cpu().numpy(), demo[name + "/nn3_idx"]), "three_nn ids differ"
    assert np.array_equal(lv["nn3_dist"][0].cpu().numpy(), demo[name + "/nn3_dist"]), "three_nn distances differ"
    got = outs["xyz_feat_att"][0].cpu().numpy()
    stride = int(demo["row_stride"])
    err = float(np.abs(got[::stride] - demo[name + "/rows"]).max())
    assert err < 1e-4, err
    cs = got.astype(np.float64).sum(0)
    assert np.abs(cs - demo[name + "/colsum"]).max() <= 1e-4 * N ** 0.5 + 1e-5 * np.abs(demo[name + "/colsum"]).max()
    # the reference's own route above 8192 points: indices as an input (core/model.py:148-155)
    outs2 = _forward(m, c[None], dev, knn_inds=torch.from_numpy(demo[name + "/knn"][None]).to(dev))
    assert torch.equal(outs2["xyz_feat_att"], outs["xyz_feat_att"])


@pytest.mark.gpu
def test_nms_keypoints_on_demo_cloud(dev, demo):
    """--perform_nms (localdesc_extract.py:92-102): keypoints of 268.bin from the HIP detector scores, against the
    float64 restatement of core/utils.py:15-43 on the same scores."""
    from scipy.spatial import cKDTree
    from dh3d_amd import utils
    c = demo["local_268"]
    m, _ = _weights("detection_config", demo)
    m.config.num_points = len(c)
    m = m.to(dev).eval().prepare()
    res = _forward(m, c[None], dev)["xyz_feat_att"][0]
    xyz, att = res[:, 0:3].contiguous(), (1 - res[:, -1]).contiguous()
    num, idx = utils.single_nms(xyz, att, nms_radius=0.5, min_response_ratio=0.01, max_keypoints=512)
    a = att.cpu().numpy().copy()
    x64 = xyz.cpu().numpy().astype(np.float64)
    dist, ind = cKDTree(x64).query(x64, k=50)
    a[dist[:, 7] > 2.0] = 0.0
    ka = a[ind]
    ka[dist > 0.5] = 0.0
    is_max = np.where(np.argmax(ka, axis=1) == 0)[0]
    thr = a.max() * 0.01
    exp = [j for _, j in sorted([(a[j], j) for j in is_max if a[j] > thr], reverse=True)][:512]
    got = idx.cpu().tolist()
    # (float32 kNN distances against the ball tree's float64: a neighbour exactly on the 0.5 m / 2.0 m shell may flip)
    assert num == len(got) and len(set(got) ^ set(exp)) <= max(2, len(exp) // 100), (num, len(exp))


@pytest.mark.gpu
def test_global_demo_clouds(dev, demo):
    """globaldesc_extract.py:61-119: global_config, two demo clouds in one batch at N = 4096 (+ the reference's zero
    padding of the last batch: an all-zero cloud rides along and must not disturb its neighbours), one at N = 8192."""
    m, _ = _weights("global_config", demo)
    m = m.to(dev).eval().prepare()
    batch = np.stack([demo["global_a"], demo["global_b"], np.zeros((4096, 3), np.float32)])
    outs = _forward(m, batch, dev)
    for b, name in enumerate(("global_a", "global_b")):
        assert np.array_equal(outs["knn_inds"][b].cpu().numpy(), demo[name + "/knn"]), name
        assert np.array_equal(m._last_geo._lv["idx"][b].cpu().numpy(), demo[name + "/fps_idx"]), name
        err = float(np.abs(outs["globaldesc"][b].cpu().numpy() - demo[name + "/globaldesc"]).max())
        assert err < 1e-4, (name, err)
    assert np.isfinite(outs["globaldesc"][2].cpu().numpy()).all()
    m2, _ = _weights("global_config", demo)
    m2.config.num_points = 8192
    m2 = m2.to(dev).eval().prepare()
    o = _forward(m2, demo["global_c"][None], dev)
    assert np.array_equal(o["knn_inds"][0].cpu().numpy(), demo["global_c/knn"])
    err = float(np.abs(o["globaldesc"][0].cpu().numpy() - demo["global_c/globaldesc"]).max())
    assert err < 1e-4, err

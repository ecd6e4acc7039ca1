"""Helpers either side of the hot path (dh3d_amd/utils.py): .bin round trips, fixed-size clouds (CPU) and keypoint
NMS against a float64 restatement of core/utils.py:15-43 (GPU)."""
import numpy as np
import pytest
import torch


def test_bin_round_trip(tmp_path):
    from dh3d_amd import utils
    rng = np.random.default_rng(0)
    pts = rng.random((100, 3), dtype=np.float32)
    utils.write_to_bin(pts, str(tmp_path / "c.bin"))
    assert np.array_equal(utils.load_single_pcfile(str(tmp_path / "c.bin")), pts)
    desc = rng.random((7, 131), dtype=np.float32)
    utils.write_to_bin(desc, str(tmp_path / "d.bin"))
    assert np.array_equal(utils.load_descriptor_bin(str(tmp_path / "d.bin")), desc)
    pts4 = rng.random((10, 4), dtype=np.float32)  # x y z intensity
    utils.write_to_bin(pts4, str(tmp_path / "e.bin"))
    assert np.array_equal(utils.load_single_pcfile(str(tmp_path / "e.bin"), dim=4), pts4[:, :3])


def test_fixednum_pcd():
    from dh3d_amd import utils
    rng = np.random.default_rng(1)
    cloud = rng.standard_normal((500, 3)).astype(np.float32)
    c, kept = utils.get_fixednum_pcd(cloud, 200, rng=np.random.default_rng(2))
    assert c.shape == (200, 3) and kept == 200
    d = ((cloud - cloud.mean(0)) ** 2).sum(1)
    inner = set(map(tuple, cloud[np.argsort(d)[:200]]))
    assert set(map(tuple, c)) == inner  # the 200 points nearest to the centroid, permuted
    c, kept = utils.get_fixednum_pcd(cloud, 800, rng=np.random.default_rng(3))
    assert c.shape == (800, 3) and kept == 500 and np.array_equal(c[:500], cloud)
    assert set(map(tuple, c[500:])) <= set(map(tuple, cloud))
    c, kept = utils.get_fixednum_pcd(cloud, 600, randsample=False)
    assert np.all(c[500:] == 100000.0)


@pytest.mark.gpu
def test_single_nms_vs_float64_restatement(dev):
    from scipy.spatial import cKDTree
    from dh3d_amd import utils
    rng = np.random.default_rng(7)
    N = 3000
    xyz = (rng.random((N, 3)) * 12).astype(np.float32)
    xyz[:40] += 100.0  # sparse outliers: muted by remove_noise
    att = rng.random(N).astype(np.float32)
    num, idx = utils.single_nms(torch.from_numpy(xyz).to(dev), torch.from_numpy(att).to(dev), nms_radius=0.8,
                                min_response_ratio=0.05, max_keypoints=256)
    # restatement of core/utils.py:15-43 with an exact float64 50-NN
    dist, ind = cKDTree(xyz.astype(np.float64)).query(xyz.astype(np.float64), k=50)
    a = att.copy()
    a[dist[:, 7] > 2.0] = 0.0
    ka = a[ind]
    ka[dist > 0.8] = 0.0
    is_max = np.where(np.argmax(ka, axis=1) == 0)[0]
    thr = a.max() * 0.05
    exp = [m for _, m in sorted([(a[m], m) for m in is_max if a[m] > thr], reverse=True)][:256]
    assert num == len(exp) and idx.cpu().tolist() == exp
    assert not (set(exp) & set(range(40)))


def test_desc_local_loss_vs_numpy():
    from dh3d_amd import losses
    rng = np.random.default_rng(3)
    B, M, D = 2, 40, 16
    xyz0 = rng.random((B, M, 3)) * 3
    R = np.stack([np.linalg.qr(rng.standard_normal((3, 3)))[0] for _ in range(B)])
    xyz1 = xyz0 @ R + rng.standard_normal((B, M, 3)) * 0.2
    f = rng.standard_normal((2 * B, M, D))
    f /= np.linalg.norm(f, axis=-1, keepdims=True)
    outs = {"xyz_sampled": torch.tensor(np.concatenate([xyz0, xyz1])), "feat_sampled": torch.tensor(f), "R": torch.tensor(R)}
    got = float(losses.desc_local_loss(outs, pos_r=0.5, search_r=20, margin=0.8, neg_weight=5))
    dx = np.sqrt(((xyz0 @ R)[:, :, None] - xyz1[:, None]) ** 2).sum(-1) if False else \
        np.sqrt((((xyz0 @ R)[:, :, None] - xyz1[:, None]) ** 2).sum(-1) + 1e-10)
    df = np.sqrt(((f[:B][:, :, None] - f[B:][:, None]) ** 2).sum(-1) + 1e-10)
    pos, neg = dx < 0.5, (dx > 1.0) & (dx < 20)
    exp = (pos * df).sum() / (pos.sum() + 1e-10) + 5 * (neg * np.maximum(0.8 - df, 0)).sum() / (neg.sum() + 1e-10)
    assert abs(got - exp) < 1e-9 * max(1, abs(exp))


@pytest.mark.gpu
def test_local_detection_loss_vs_numpy(dev):
    from scipy.spatial import cKDTree
    from dh3d_amd import losses
    rng = np.random.default_rng(4)
    B, N, M, D, k = 2, 300, 32, 8, 16
    xyz0 = (rng.random((B, N, 3)) * 4).astype(np.float32)
    R = np.stack([np.linalg.qr(rng.standard_normal((3, 3)))[0] for _ in range(B)]).astype(np.float32)
    xyz1 = (xyz0 @ R + rng.standard_normal((B, N, 3)) * 0.05).astype(np.float32)
    feat = rng.standard_normal((2 * B, N, D)).astype(np.float32)
    samp = np.stack([rng.choice(N, M, replace=False) for _ in range(2 * B)])[..., None].astype(np.int32)
    xyz = np.concatenate([xyz0, xyz1])
    xs = np.take_along_axis(xyz, samp.astype(np.int64), 1)
    fs = np.take_along_axis(feat, samp.astype(np.int64), 1)
    att = rng.random((2 * B, M, 1)).astype(np.float32)
    outs = {k_: torch.from_numpy(v).to(dev) for k_, v in dict(xyz=xyz, feat=feat, sample_nodes_concat=samp,
            att_sampled=att, xyz_sampled=xs, feat_sampled=fs, R=R).items()}
    got = float(losses.local_detection_loss_nn(outs, det_k=k))
    # numpy restatement of core/losses.py:66-133
    tot = []
    for b in range(B):
        knn = cKDTree(xyz1[b].astype(np.float64)).query(xyz1[b].astype(np.float64), k=k)[1]
        warp = xs[b] @ R[b]
        cand = knn[samp[B + b, :, 0]]
        dneg = np.sqrt(((fs[b][:, None] - fs[B + b][None]) ** 2).sum(-1) + 1e-10) + \
            (1 - (np.sqrt(((warp[:, None] - xs[B + b][None]) ** 2).sum(-1) + 1e-10) > 1)) * 100
        cand = np.concatenate([cand, knn[dneg.argmin(1)]], 1)
        dx = np.sqrt(((warp[:, None] - xyz1[b][cand]) ** 2).sum(-1))
        df = ((fs[b][:, None] - feat[B + b][cand]) ** 2).sum(-1)
        order = np.argsort(df, axis=1, kind="stable")[:, :5]
        good = np.concatenate([np.take_along_axis(dx, order, 1) <= 0.3, np.ones((M, 1), bool)], 1)
        AR = (good.argmax(1) + 1e-8) / 5
        s0 = att[b, :, 0]
        tot.append(1 - (AR * s0 + 0.3 * (1 - s0)))
    assert abs(got - float(np.mean(tot))) < 1e-5

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

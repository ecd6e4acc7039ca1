"""Retrieval metrics against the reference's recipe restated with scipy (evaluation_retrieval.py:25-61)."""
import numpy as np
import torch
from scipy.spatial import cKDTree


def _ref_metrics(ref_d, qry_d, gt, k):
    _, ind = cKDTree(ref_d).query(qry_d, k=k)
    thr = max(int(round(len(ref_d) / 100.0)), 1)
    tp = gt[np.arange(len(ind))[:, None], ind]
    valid = gt.any(1)
    return (np.cumsum(tp, 1)[valid] > 0).mean(0), tp[:, :thr].any(1)[valid].mean()


def test_recall_matches_kdtree_recipe():
    from dh3d_amd import evaluation as ev
    rng = np.random.default_rng(3)
    R, Q, D, k = 700, 300, 32, 25
    ref_pos = rng.random((R, 2)) * 2000
    qry_pos = ref_pos[rng.integers(0, R, Q)] + rng.normal(0, 20, (Q, 2))
    qry_pos[:17] += 1e5  # queries with no true match are excluded from the averages
    ref_d = rng.standard_normal((R, D)).astype(np.float32)
    # descriptors correlated with place: some retrievals succeed, some fail
    nearest = np.argmin(((qry_pos[:, None] - ref_pos[None]) ** 2).sum(2), 1)
    qry_d = (ref_d[nearest] + 0.9 * rng.standard_normal((Q, D))).astype(np.float32)
    gt = np.linalg.norm(qry_pos[:, None] - ref_pos[None], axis=2) < 25
    assert np.array_equal(ev.is_gt_match_2d(qry_pos, ref_pos, 25).numpy(), gt)
    assert np.array_equal(ev.retrieval(ref_d, qry_d, k).numpy(), cKDTree(ref_d).query(qry_d, k=k)[1])
    rec, one = ev.evaluate_pair(ref_d, ref_pos, qry_d, qry_pos, max_num_nn=k)
    rec0, one0 = _ref_metrics(ref_d, qry_d, gt, k)
    assert np.allclose(rec.numpy(), rec0) and abs(one - one0) < 1e-12
    assert 0.2 < rec0[0] < 0.99 and rec0[-1] >= rec0[0]  # a meaningful case, monotone in N
    out = ev.evaluate_sets([("a", ref_d, ref_pos), ("b", qry_d, qry_pos)], [("a", ref_d, ref_pos), ("b", qry_d, qry_pos)],
                           max_num_nn=k)
    assert [(p[0], p[1]) for p in out["pairs"]] == [("a", "b"), ("b", "a")]  # same-name pairs skipped (:140)
    assert torch.allclose(out["avg_recall"], (out["pairs"][0][2] + out["pairs"][1][2]) / 2)


def test_no_valid_query_is_nan():
    from dh3d_amd import evaluation as ev
    r = np.zeros((5, 4), np.float32)
    rec, one = ev.evaluate_pair(r, np.zeros((5, 2)), r, np.full((5, 2), 1e6), max_num_nn=3)
    assert rec.shape == (3,) and bool(torch.isnan(rec).all()) and one != one


import pytest


@pytest.mark.gpu
def test_retrieval_metrics_on_device_descriptors_vs_kdtree_recipe(dev):
    """The evaluation as globaldesc_extract.py + evaluation_retrieval.py run it, end to end on the device: global
    descriptors of two synthetic 'traversals' (the same places, re-sampled and jittered) out of the HIP forward, metrics
    computed on those DEVICE tensors, against the reference's recipe (scipy cKDTree on host copies,
    evaluation_retrieval.py:37-53,129-169)."""
    from dh3d_amd import ConfigFactory, evaluation as ev
    from dh3d_amd.model import DH3D
    m = DH3D(ConfigFactory("global_config").getconfig()).init_synthetic(7).to(dev).eval().prepare()
    rng = np.random.default_rng(11)
    places, N = 48, 2048
    base = rng.random((places, N, 3), dtype=np.float32) * np.array([40, 40, 6], np.float32)
    base *= (0.5 + rng.random((places, 1, 3))).astype(np.float32)  # every place its own extent
    for p in range(places):  # every place its own structure: a few dense blobs
        c = rng.integers(0, N, 5)
        base[p, : N // 2] = base[p, c[rng.integers(0, 5, N // 2)]] + rng.normal(0, 1.0 + 0.1 * p, (N // 2, 3)).astype(np.float32)
    ref_pos = rng.random((places, 2)) * 3000
    qsel = rng.permutation(places)[:40]
    qry = base[qsel][:, rng.permutation(N)] + rng.normal(0, 0.05, (40, N, 3)).astype(np.float32)
    qry_pos = ref_pos[qsel] + rng.normal(0, 8, (40, 2))
    qry_pos[:3] += 1e5  # three queries without a true match
    with torch.no_grad():
        ref_d = m(torch.from_numpy(base).to(dev), fetch=("globaldesc",))["globaldesc"]
        qry_d = m(torch.from_numpy(qry).to(dev), fetch=("globaldesc",))["globaldesc"]
    assert ref_d.is_cuda and ref_d.shape == (places, 256)
    k = 25
    gt_dev = ev.is_gt_match_2d(torch.from_numpy(qry_pos).to(dev), torch.from_numpy(ref_pos).to(dev), 25.0)
    idx = ev.retrieval(ref_d, qry_d, k)
    assert idx.is_cuda and gt_dev.is_cuda
    rec, one = ev.evaluate_pair(ref_d, torch.from_numpy(ref_pos).to(dev), qry_d, torch.from_numpy(qry_pos).to(dev), max_num_nn=k)
    # the reference's recipe on host copies of the SAME descriptors
    rd, qd = ref_d.cpu().numpy(), qry_d.cpu().numpy()
    gt = np.linalg.norm(qry_pos[:, None] - ref_pos[None], axis=2) < 25
    assert np.array_equal(gt_dev.cpu().numpy(), gt)
    dist, ind = cKDTree(rd).query(qd, k=k)
    got = idx.cpu().numpy()
    same = got == ind
    if not same.all():  # only exact ties of the float32 descriptors' distances may order differently
        q, j = np.nonzero(~same)
        dg = np.linalg.norm(qd[q].astype(np.float64) - rd[got[q, j]].astype(np.float64), axis=1)
        assert np.allclose(dg, dist[q, j], rtol=0, atol=1e-12)
    rec0, one0 = _ref_metrics(rd, qd, gt, k)
    assert np.allclose(rec.cpu().numpy(), rec0) and abs(one - one0) < 1e-12
    # a meaningful case even with random-init weights: well above chance (1 / places), monotone in N, not saturated
    assert rec0[0] > 3.0 / places and rec0[-1] >= rec0[0] and rec0[0] < 1.0

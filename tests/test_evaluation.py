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

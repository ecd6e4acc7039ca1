"""numpy restatement of the DH3D losses (TEST INFRASTRUCTURE ONLY -- never imported by dh3d_amd).

Follows core/losses.py statement by statement, float32 like the TF graph:
    desc_local_loss          :29-63      local_detection_loss_nn :66-133
    best_pos_distance        :137-145    lazy_triplet_loss_impl  :147-160
    lazy_triplet_loss        :163-170    lazy_quadruplet_loss    :173-200
    pairwise_dist            core/tf_utils.py:125-136
PARITY UNPINNED: the reference has no test and no stored vector for any loss; these restatements are
anchored on the source text only.  The kNN of the detector loss is the C oracle's (knn_bruteforce).
"""
import numpy as np

from . import cpu as O

F = np.float32


def pairwise_dist(A, B):
    """[b,n,d], [b,m,d] -> [b,n,m] sum of squared differences (tf_utils.py:125-136)."""
    d = A[:, :, None, :] - B[:, None, :, :]
    return np.sum(d * d, axis=3, dtype=F)


def desc_local_loss(outs, pos_r=0.5, search_r=20, margin=0.8, neg_weight=5, **kwargs):
    xyz0, xyz1 = np.split(outs["xyz_sampled"], 2, axis=0)
    feat0, feat1 = np.split(outs["feat_sampled"], 2, axis=0)
    xyz0_warp = np.matmul(xyz0, outs["R"]).astype(F)
    xyzdist = np.sqrt(pairwise_dist(xyz0_warp, xyz1) + F(1e-10))
    is_neg = np.logical_and(xyzdist > F(pos_r * 2), xyzdist < F(search_r)).astype(F)
    is_pos = (xyzdist < F(pos_r)).astype(F)
    feat_dist = np.sqrt(pairwise_dist(feat0, feat1) + F(1e-10))
    num_pos = F(np.count_nonzero(is_pos))
    num_neg = F(np.count_nonzero(is_neg))
    pos_loss = np.sum(is_pos * feat_dist, dtype=F) / (num_pos + F(1e-10))
    neg_loss = np.sum(is_neg * np.maximum(F(margin) - feat_dist, 0), dtype=F) / (num_neg + F(1e-10))
    return F(pos_loss + F(neg_weight) * neg_loss)


def local_detection_loss_nn(outs, ar_th=0.3, det_k=16, ar_nn_k=5, pos_r=0.3, use_hardest_neg=True, **unused):
    xyz0, xyz1 = np.split(outs["xyz"], 2, axis=0)
    _, feat1 = np.split(outs["feat"], 2, axis=0)
    _, sample_ind1 = np.split(outs["sample_nodes_concat"], 2, axis=0)
    score0, _ = np.split(outs["att_sampled"], 2, axis=0)
    xyz_s0, xyz_s1 = np.split(outs["xyz_sampled"], 2, axis=0)
    feat_s0, feat_s1 = np.split(outs["feat_sampled"], 2, axis=0)
    knn1, _ = O.knn_bruteforce(np.ascontiguousarray(xyz1.transpose(0, 2, 1)), det_k)     # op output [B,N,K]
    B, M = xyz_s0.shape[0], xyz_s0.shape[1]
    xyz0_warp = np.matmul(xyz_s0, outs["R"]).astype(F)
    bidx = np.arange(B).reshape(B, 1)
    knn_sampled1 = knn1[bidx, sample_ind1.reshape(B, M)]                                   # [B,M,k]
    if use_hardest_neg:
        d_all = np.sqrt(pairwise_dist(xyz0_warp, xyz_s1) + F(1e-10))
        is_neg = (d_all > 1).astype(F)
        feat_dist_all = np.sqrt(pairwise_dist(feat_s0, feat_s1) + F(1e-10))
        neg_dist = feat_dist_all + (1 - is_neg) * F(100)
        hardest = np.argmin(neg_dist, axis=2)                                              # [B,M]
        knn_sampled1 = np.concatenate([knn_sampled1, knn1[bidx, hardest]], -1)
    b3 = np.arange(B).reshape(B, 1, 1)
    sampled_xyz1 = xyz1[b3, knn_sampled1]
    sampled_feat1 = feat1[b3, knn_sampled1]
    dx = xyz0_warp[:, :, None, :] - sampled_xyz1
    matching_xyz_dist = np.sqrt(np.sum(dx * dx, -1, dtype=F))
    df = feat_s0[:, :, None, :] - sampled_feat1
    matching_feat_dist = np.sum(df * df, -1, dtype=F)
    order = np.argsort(matching_feat_dist, axis=-1, kind="stable")[..., :5]                # tf.nn.top_k(-d, k=5) (:114)
    sel = np.take_along_axis(matching_xyz_dist, order, axis=2)
    is_good = (sel <= F(pos_r)).astype(F)
    is_good = np.concatenate([is_good, np.ones(is_good.shape[:2] + (1,), F)], -1)
    first = np.argmax(is_good, axis=-1).astype(F)
    AR = ((first + F(1e-8)) / F(ar_nn_k)).astype(F)
    s0 = score0[:, :, 0]
    return F(np.mean(1 - (AR * s0 + F(ar_th) * (1 - s0)), dtype=F))


def best_pos_distance(query, pos_vecs):
    d = pos_vecs - query
    return np.min(np.sum(d * d, 2, dtype=F), 1)


def lazy_triplet_loss_impl(q_vec, pos_vecs, neg_vecs, margin):
    best_pos = best_pos_distance(q_vec, pos_vecs).reshape(-1, 1)
    d = neg_vecs - q_vec
    t = np.maximum(F(margin) + (best_pos - np.sum(d * d, 2, dtype=F)), 0)
    return F(np.mean(np.max(t, 1), dtype=F))


def lazy_triplet_loss(global_descs, batch_size, num_pos, num_neg, global_triplet_margin=0.5, **kwargs):
    D = global_descs.shape[-1]
    q, p, n = _split3(global_descs, batch_size, num_pos, num_neg)
    return lazy_triplet_loss_impl(q.reshape(batch_size, 1, D), p.reshape(batch_size, num_pos, D),
                                  n.reshape(batch_size, num_neg, D), global_triplet_margin)


def _split3(g, batch_size, num_pos, num_neg):
    a, b = batch_size, batch_size + num_pos * batch_size
    c = b + num_neg * batch_size
    assert g.shape[0] == c, "tf.split sizes must sum to the row count"
    return g[:a], g[a:b], g[b:c]


def lazy_quadruplet_loss(global_descs, batch_size, num_pos, num_neg, global_triplet_margin=0.5,
                         global_quadruplet_margin=0.2, **kwargs):
    D = global_descs.shape[-1]
    a, b = batch_size, batch_size + num_pos * batch_size
    c = b + num_neg * batch_size
    assert global_descs.shape[0] == c + batch_size, "tf.split sizes must sum to the row count"
    q = global_descs[:a].reshape(batch_size, 1, D)
    pos = global_descs[a:b].reshape(batch_size, num_pos, D)
    neg = global_descs[b:c].reshape(batch_size, num_neg, D)
    oth = global_descs[c:].reshape(batch_size, 1, D)
    trip = lazy_triplet_loss_impl(q, pos, neg, global_triplet_margin)
    best_pos = best_pos_distance(q, pos).reshape(-1, 1)
    d = neg - oth
    second = np.maximum(F(global_quadruplet_margin) + (best_pos - np.sum(d * d, 2, dtype=F)), 0)
    return F(trip + F(np.mean(np.max(second, 1), dtype=F)))

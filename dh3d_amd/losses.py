"""Global-descriptor losses (mirrors core/losses.py:137-200): lazy triplet / quadruplet.

Plain tensor math on [Bt, 256] descriptors -- not a hot kernel; it defines what the multi-GPU
all-gather must deliver and in which order: rows are role-ordered
[anchors (B), positives (P*B), negatives (Ng*B), other-negatives (B)]  (core/losses.py:175-178).
"""
import torch


def best_pos_distance(query, pos_vecs):
    """query [B,1,D], pos_vecs [B,P,D] -> min_p ||q - p||^2  (core/losses.py:137-145)."""
    return ((pos_vecs - query) ** 2).sum(2).min(1).values


def lazy_triplet_loss_impl(q_vec, pos_vecs, neg_vecs, margin):
    """mean_b max_j relu(margin + best_pos - ||q - n_j||^2)  (core/losses.py:147-160)."""
    best_pos = best_pos_distance(q_vec, pos_vecs).reshape(-1, 1)
    neg_d = ((neg_vecs - q_vec) ** 2).sum(2)
    return torch.clamp(margin + best_pos - neg_d, min=0).max(1).values.mean()


def _split(global_descs, batch_size, num_pos, num_neg, other_neg):
    sizes = [batch_size, num_pos * batch_size, num_neg * batch_size] + ([batch_size] if other_neg else [])
    if sum(sizes) != global_descs.shape[0]:
        raise ValueError("descriptor rows %d do not match role sizes %s" % (global_descs.shape[0], sizes))
    parts = torch.split(global_descs, sizes, dim=0)
    D = global_descs.shape[-1]
    q = parts[0].reshape(batch_size, 1, D)
    pos = parts[1].reshape(batch_size, num_pos, D)
    neg = parts[2].reshape(batch_size, num_neg, D)
    other = parts[3].reshape(batch_size, 1, D) if other_neg else None
    return q, pos, neg, other


def lazy_triplet_loss(global_descs, batch_size, num_pos, num_neg, global_triplet_margin=0.5, **kwargs):
    q, pos, neg, _ = _split(global_descs, batch_size, num_pos, num_neg, False)
    return lazy_triplet_loss_impl(q, pos, neg, global_triplet_margin)


def lazy_quadruplet_loss(global_descs, batch_size, num_pos, num_neg, global_triplet_margin=0.5,
                         global_quadruplet_margin=0.2, **kwargs):
    """core/losses.py:173-200."""
    q, pos, neg, other = _split(global_descs, batch_size, num_pos, num_neg, True)
    trip = lazy_triplet_loss_impl(q, pos, neg, global_triplet_margin)
    best_pos = best_pos_distance(q, pos).reshape(-1, 1)
    d_other = ((neg - other) ** 2).sum(2)
    second = torch.clamp(global_quadruplet_margin + best_pos - d_other, min=0).max(1).values.mean()
    return trip + second


# ---------------------------------------------------------------------------------------------------------
# Local-descriptor and detector losses (core/losses.py:29-133): plain tensor math on the sampled keypoints of a
# registered cloud pair [cloud 0 batch | cloud 1 batch]; the 16-NN of the detector loss runs on the kNN kernel.
def _pairwise_sqdist(a, b):
    """[B,n,D], [B,m,D] -> [B,n,m] sum of squared differences (core/tf_utils.py:125-136).  float32 GPU descriptors go
    through the HIP kernel (train_ops.pairwise_sqdist: the broadcast form materialises [B,n,m,D] -- 1.3 GB and a third of
    the stage-1 training step at 10 x 512 x 512 x 128); anything else (float64 test restatements, coordinates) stays plain
    tensor math."""
    if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape[2] >= 16 and a.shape[2] % 4 == 0:
        from . import train_ops as T
        return T.pairwise_sqdist(a, b)
    return ((a.unsqueeze(2) - b.unsqueeze(1)) ** 2).sum(3)


def desc_local_loss(outs, pos_r=0.5, search_r=20, margin=0.8, neg_weight=5, **unused):
    """n-tuple loss on sampled keypoints (core/losses.py:29-63).  outs: 'xyz_sampled' [2B,M,3], 'feat_sampled'
    [2B,M,D], 'R' [B,3,3] (cloud 0 -> cloud 1).  Returns pos_loss + neg_weight * neg_loss."""
    xyz0, xyz1 = torch.chunk(outs["xyz_sampled"], 2, dim=0)
    f0, f1 = torch.chunk(outs["feat_sampled"], 2, dim=0)
    d_xyz = torch.sqrt(_pairwise_sqdist(torch.matmul(xyz0, outs["R"]), xyz1) + 1e-10)
    is_neg = ((d_xyz > pos_r * 2) & (d_xyz < search_r)).to(f0.dtype)
    is_pos = (d_xyz < pos_r).to(f0.dtype)
    d_feat = torch.sqrt(_pairwise_sqdist(f0, f1) + 1e-10)
    pos_loss = (is_pos * d_feat).sum() / (is_pos.sum() + 1e-10)
    neg_loss = (is_neg * torch.clamp(margin - d_feat, min=0)).sum() / (is_neg.sum() + 1e-10)
    return pos_loss + neg_weight * neg_loss


def local_detection_loss_nn(outs, ar_th=0.3, det_k=16, ar_nn_k=5, pos_r=0.3, use_hardest_neg=True, **unused):
    """Detector loss (core/losses.py:66-133): for every sampled keypoint of cloud 0, the rank (among its `ar_nn_k`
    closest-in-feature candidates) of the first candidate that lies within pos_r of the warped keypoint; candidates =
    the det_k nearest neighbours in cloud 1 of the corresponding sample (plus those of the hardest negative).
    outs: 'xyz' [2B,N,3], 'feat' [2B,N,D], 'sample_nodes_concat' [2B,M,1] int, 'att_sampled' [2B,M,1],
    'xyz_sampled', 'feat_sampled', 'R'."""
    from . import ops
    xyz0, xyz1 = torch.chunk(outs["xyz"], 2, dim=0)
    _, feat1 = torch.chunk(outs["feat"], 2, dim=0)
    _, samp1 = torch.chunk(outs["sample_nodes_concat"], 2, dim=0)
    score0, _ = torch.chunk(outs["att_sampled"], 2, dim=0)
    xyz_s0, xyz_s1 = torch.chunk(outs["xyz_sampled"], 2, dim=0)
    feat_s0, feat_s1 = torch.chunk(outs["feat_sampled"], 2, dim=0)
    B, M = xyz_s0.shape[0], xyz_s0.shape[1]
    knn1, _ = ops.knn_bruteforce(xyz1.transpose(1, 2).contiguous(), k=det_k)  # [B,N,k]
    knn1 = knn1.long()
    xyz0_warp = torch.matmul(xyz_s0, outs["R"])
    bidx = torch.arange(B, device=xyz0.device).reshape(B, 1)
    cand = knn1[bidx, samp1.reshape(B, M).long()]                           # [B,M,k]
    if use_hardest_neg:
        is_neg = (torch.sqrt(_pairwise_sqdist(xyz0_warp, xyz_s1) + 1e-10) > 1).to(feat_s0.dtype)
        neg_dist = torch.sqrt(_pairwise_sqdist(feat_s0, feat_s1) + 1e-10) + (1 - is_neg) * 100
        hardest = neg_dist.argmin(dim=2)                                     # [B,M] -- index used as in the reference
        cand = torch.cat([cand, knn1[bidx, hardest]], dim=-1)
    b3 = torch.arange(B, device=xyz0.device).reshape(B, 1, 1)
    cxyz, cfeat = xyz1[b3, cand], feat1[b3, cand]                            # [B,M,k',3], [B,M,k',D]
    d_xyz = torch.sqrt(((xyz0_warp.unsqueeze(2) - cxyz) ** 2).sum(-1))
    d_feat = ((feat_s0.unsqueeze(2) - cfeat) ** 2).sum(-1)
    _, order = torch.topk(-d_feat, k=ar_nn_k, dim=-1)
    good = (torch.gather(d_xyz, 2, order) <= pos_r).to(feat_s0.dtype)
    good = torch.cat([good, torch.ones_like(good[..., :1])], dim=-1)
    first = good.argmax(dim=-1).to(feat_s0.dtype)                            # first index of the maximum
    AR = (first + 1e-8) / ar_nn_k
    s0 = score0.squeeze(2)
    return (1 - (AR * s0 + ar_th * (1 - s0))).mean()


def compute_loss(outs, config):
    """The reference's loss assembly (core/model.py:212-237): every loss is called with **config (so the config's
    margin / pos_r / ar_th / ... override the function defaults, as upstream) and scaled by its *_loss_weight.
    outs: the forward's named outputs (+ the sampled-keypoint tensors the local losses need); the global descriptor
    is read from 'global_desc' (the reference's key, model.py:206) or 'globaldesc' (the fetch name).
    Weight decay on '.*/W' (model.py:239-243) is the optimiser's part: training.QuadrupletTrainer."""
    import sys
    mod = sys.modules[__name__]

    def weight(key):
        return config.get(key) if config.get(key) is not None else 1.0
    total = 0.0
    if config.get("extract_global"):
        desc = outs["global_desc"] if "global_desc" in outs else outs["globaldesc"]
        fn = getattr(mod, config.get("global_loss") or "lazy_quadruplet_loss")
        total = total + fn(global_descs=desc, **config) * weight("global_loss_weight")
    if config.get("add_local_loss"):
        fn = getattr(mod, config.get("local_loss") or "desc_local_loss")
        total = total + fn(outs, **config) * weight("local_loss_weight")
    if config.get("detection") and config.get("add_det_loss"):
        fn = getattr(mod, config.get("detection_loss") or "local_detection_loss_nn")
        total = total + fn(outs, **config) * weight("det_loss_weight")
    return total

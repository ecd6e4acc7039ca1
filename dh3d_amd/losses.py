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

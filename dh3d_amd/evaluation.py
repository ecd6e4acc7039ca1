"""Place-recognition retrieval metrics (evaluate/global_eval/evaluation_retrieval.py:25-61,129-169): recall@N and the
top-1 % recall of global descriptors between traversals.  Tensor in, tensor out, on whatever device the descriptors
live (the reference builds a scipy cKDTree on the host); no file-system conventions."""
import torch


def is_gt_match_2d(query_pos, ref_pos, distance_thresh=25.0):
    """[Q,2] (northing, easting), [R,2] -> bool [Q,R]: reference within `distance_thresh` metres of the query
    (evaluation_retrieval.py:25-34)."""
    q = torch.as_tensor(query_pos, dtype=torch.float64)
    r = torch.as_tensor(ref_pos, dtype=torch.float64).to(q.device)
    return torch.cdist(q, r) < distance_thresh


def retrieval(ref_descriptors, query_descriptors, max_num_nn):
    """Indices [Q, max_num_nn] of the nearest reference descriptors, nearest first (:37-40; Euclidean, exact)."""
    ref = torch.as_tensor(ref_descriptors, dtype=torch.float32)
    qry = torch.as_tensor(query_descriptors, dtype=torch.float32).to(ref.device)
    k = min(int(max_num_nn), ref.shape[0])
    out = torch.empty((qry.shape[0], k), dtype=torch.int64, device=ref.device)
    for s in range(0, qry.shape[0], 4096):  # bounded [chunk, R] distance matrix
        d = torch.cdist(qry[s:s + 4096].double(), ref.double())  # float64: ranks as the k-d tree's exact distances
        out[s:s + 4096] = torch.topk(d, k, dim=1, largest=False, sorted=True).indices
    return out


def compute_tp_fp(ref_descriptors, query_descriptors, gt_matches, max_num_nn=25):
    """(tp_cum [Q,k], fp_cum [Q,k], valid [Q], one_percent_retrieved [Q]) as the reference's function of the same name
    (:43-54): a query is `valid` if it has any true match; `one_percent` looks at the first max(round(R/100), 1)."""
    gt = torch.as_tensor(gt_matches, dtype=torch.bool)
    idx = retrieval(ref_descriptors, query_descriptors, max_num_nn).to(gt.device)
    threshold = max(int(round(gt.shape[1] / 100.0)), 1)
    tp = torch.gather(gt, 1, idx)
    return tp.cumsum(1), (~tp).cumsum(1), gt.any(1), tp[:, :threshold].any(1)


def evaluate_pair(ref_desc, ref_pos, query_desc, query_pos, max_num_nn=25, distance_thresh=25.0):
    """recall@1..max_num_nn [k] and top-1 % recall (scalar) of one (database traversal, query traversal) pair (:143-153)."""
    gt = is_gt_match_2d(query_pos, ref_pos, distance_thresh)
    tp, _, valid, one = compute_tp_fp(ref_desc, query_desc, gt, max_num_nn)
    if not bool(valid.any()):
        nan = float("nan")
        return torch.full((tp.shape[1],), nan), nan
    recall = (tp[valid] > 0).double().mean(0)
    return recall, float(one[valid].double().mean())


def evaluate_sets(database, queries, max_num_nn=25, distance_thresh=25.0):
    """database / queries: lists of (name, descriptors [n,D], positions [n,2]).  Every database traversal against every
    query traversal of a different name (:139-142); returns {"pairs": [(ref, query, recalls, one_percent)],
    "avg_recall": [k], "avg_one_percent_retrieved": float} (:159-169)."""
    pairs = []
    for rname, rdesc, rpos in database:
        for qname, qdesc, qpos in queries:
            if rname == qname:
                continue
            rec, one = evaluate_pair(rdesc, rpos, qdesc, qpos, max_num_nn, distance_thresh)
            pairs.append((rname, qname, rec, one))
    if not pairs:
        return {"pairs": [], "avg_recall": None, "avg_one_percent_retrieved": None}
    return {"pairs": pairs, "avg_recall": torch.stack([p[2] for p in pairs]).mean(0),
            "avg_one_percent_retrieved": float(sum(p[3] for p in pairs) / len(pairs))}

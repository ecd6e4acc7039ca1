"""Host-side helpers either side of the hot path: keypoint NMS on the device, raw .bin clouds and
descriptors, fixed-size clouds.  Mirrors core/utils.py:15-43 (single_nms), :87-110 (get_fixednum_pcd),
:139-153 (load_descriptor_bin / load_single_pcfile / write_to_bin)."""
import numpy as np
import torch

from . import pm


def single_nms(xyz, attention, nms_radius, min_response_ratio, max_keypoints, remove_noise=True, knn=50):
    """Keypoint non-maximum suppression on the GPU (core/utils.py:15-43, used by localdesc_extract.py:92-102).

    xyz [N,3] / attention [N] CUDA float32.  A point survives if its response is the (first) maximum among its `knn`
    nearest neighbours inside `nms_radius` (rank 0 = the point itself), exceeds min_response_ratio * max response,
    and -- remove_noise -- its 8th neighbour lies within 2.0 (sparse outliers are muted first).  Survivors are
    ordered by (response, index) descending and cut to max_keypoints.  Returns (num_keypoints, indices [num] int64).

    The 50-NN query is the kNN kernel of the hot path (exact; float32 distances where the reference's scikit-learn
    ball tree computes in float64 -- a point exactly on the 2.0 / nms_radius shell may fall on the other side)."""
    if xyz.dim() != 2 or xyz.shape[1] != 3:
        raise ValueError("xyz must be [N,3]")
    N = xyz.shape[0]
    k = min(knn, N)
    att = attention.reshape(N).to(torch.float32).clone()
    nn, dist = pm.knn_xyz(xyz.reshape(1, N, 3).contiguous(), k)
    nn, dist = nn[0].long(), dist[0]
    if remove_noise and k > 7:
        att[dist[:, 7] > 2.0] = 0.0
    knn_att = att[nn]
    knn_att[dist > nms_radius] = 0.0
    is_max = knn_att.argmax(dim=1) == 0  # torch.argmax, like numpy's, returns the first maximal index
    thresh = att.max() * min_response_ratio
    keep = torch.nonzero(is_max & (att > thresh)).reshape(-1)
    if keep.numel() > 0:
        # sorted(..., reverse=True) on (attention, index) tuples: response descending, then index descending
        order = torch.argsort(keep, descending=True)
        keep = keep[order]
        keep = keep[torch.argsort(att[keep], descending=True, stable=True)]
    keep = keep[:max_keypoints]
    return int(keep.numel()), keep


def load_descriptor_bin(filename, dim=131, dtype=np.float32):
    """Raw little-endian float32 rows of `dim` values: [x, y, z, 128-d descriptor(, score)]."""
    return np.fromfile(filename, dtype=dtype).reshape(-1, dim)


def load_single_pcfile(filename, dim=3, dtype=np.float32):
    """Raw float32 cloud with `dim` values per point; the first three are the coordinates."""
    pc = np.fromfile(filename, dtype=dtype)
    return pc.reshape(pc.shape[0] // dim, dim)[:, 0:3]


def write_to_bin(points, filename):
    np.ascontiguousarray(points).tofile(filename)


def get_fixednum_pcd(cloud, targetnum, randsample=True, sortby_dis=True, rng=None):
    """Crop (nearest to the centroid first, then a random permutation) or pad (random re-draws, or far-away
    dummies) a cloud to exactly `targetnum` points; returns (cloud, number of original points kept).
    The optional voxel down-sampling / outlier removal of the reference live in open3d and are out of scope."""
    rng = np.random.default_rng() if rng is None else rng
    cloud = np.asarray(cloud)
    n = cloud.shape[0]
    if n > targetnum:
        if sortby_dis:
            d = ((cloud - cloud.mean(axis=0)) ** 2).sum(axis=1)
            cloud = cloud[np.argsort(d)[:targetnum], :3]
        cloud = cloud[rng.choice(cloud.shape[0], targetnum, replace=False)]
        return cloud, targetnum
    pad = targetnum - n
    if randsample:
        extra = cloud[rng.choice(n, size=pad, replace=True)]
    else:
        extra = np.full((pad, 3), 100000.0, dtype=np.float32)
    return np.concatenate([cloud, extra], axis=0), n

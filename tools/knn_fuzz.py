"""Dev: long randomised comparison of the ordered-cloud kNN (knn_split / knn_sorted kernels: box pruning, per-query
point-to-box test, groups visited outwards, own group dealt to the waves) with the brute-force kernel -- ids AND distances
bit for bit, over many more shapes / distributions than the test-suite (uniform, blobs, planes, lines, lattices with exact
ties, duplicated points, tiny clouds, K up to 64).   python tools/knn_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from dh3d_amd import pm
dev = torch.device("cuda")
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
KINDS = ("uniform", "blobs", "plane", "line", "grid", "dups", "shell", "twoscale", "same")
def cloud(B, N, kind):
    if kind == "uniform": x = rng.random((B, N, 3))
    elif kind == "blobs":
        c = rng.random((B, 6, 3)); x = c[:, rng.integers(0, 6, N)] + 0.01 * rng.standard_normal((B, N, 3))
    elif kind == "plane": x = rng.random((B, N, 3)); x[..., 2] *= 1e-3
    elif kind == "line": x = rng.random((B, N, 1)) * np.array([1.0, 0.5, 0.25]) + 1e-4 * rng.standard_normal((B, N, 3))
    elif kind == "grid":
        g = int(np.ceil(N ** (1 / 3))) + 1
        pts = np.stack(np.meshgrid(*[np.arange(g)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
        x = np.stack([pts[rng.permutation(len(pts))[:N]] for _ in range(B)]) * 0.25
    elif kind == "dups":
        h = rng.random((B, (N + 1) // 2, 3)); x = np.concatenate([h, h], 1)[:, :N]; x = np.stack([xx[rng.permutation(N)] for xx in x])
    elif kind == "shell":
        v = rng.standard_normal((B, N, 3)); x = v / np.linalg.norm(v, axis=2, keepdims=True) * 10
    elif kind == "same": x = np.broadcast_to(rng.random((B, 1, 3)), (B, N, 3)).copy()
    else:
        x = rng.random((B, N, 3)); x[:, : N // 2] = x[:, : N // 2] * 1e-3 + 0.5
    return np.ascontiguousarray(x.astype(np.float32))
bad = 0
for it in range(cases):
    N = int(rng.choice([1, 2, 7, 63, 64, 65, 100, 513, 1000, 1024, 2049, 4096, 5000, 8192, 9999, 16384]))
    B = int(rng.choice([1, 2, 3, 8, 11, 32])) if N <= 4096 else int(rng.choice([1, 2, 4, 8]))
    K = int(rng.choice([1, 3, 4, 8, 12, 16, 17, 32, 50, 64]))
    if B * N * K > 8 * 8192 * 16: K = 8
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    t = torch.from_numpy(cloud(B, N, kind)).to(dev)
    srt, gbox = pm.spatial_sort(t)
    nn, d = pm.knn_sorted(srt, gbox, K)
    nn0, d0 = pm.knn_xyz(t, K)
    ok = bool(torch.equal(nn, nn0)) and bool(torch.equal(d.view(torch.int32), d0.view(torch.int32)))
    if not ok:
        bad += 1
        print("MISMATCH", it, B, N, K, kind, int((nn != nn0).sum()), flush=True)
print("cases %d mismatches %d" % (cases, bad))

"""Dev: long randomised comparison of the ordered-cloud FPS (candidate lists + judge) with the plain op -- many more
shapes / distributions than tests/test_stress_gpu.py (uniform, blobs, planes, lines, lattices with exact ties, duplicated
points, tiny clouds, npoint = N).   python tools/fps_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from dh3d_amd import ops, pm
dev = torch.device("cuda")
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
KINDS = ("uniform", "blobs", "plane", "line", "grid", "dups", "shell", "twoscale")
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
    else:
        x = rng.random((B, N, 3)); x[:, : N // 2] = x[:, : N // 2] * 1e-3 + 0.5
    return np.ascontiguousarray(x.astype(np.float32))
bad = 0
for it in range(cases):
    N = int(rng.choice([1, 2, 63, 64, 65, 100, 513, 1000, 1024, 2049, 4096, 5000, 8192, 9999, 12288, 13000, 16384]))
    B = int(rng.integers(1, 5)) if N > 2048 else int(rng.integers(1, 12))
    m = int(rng.choice([1, 2, max(N // 8, 1), max(N // 3, 1), N]))
    if N * m > 16384 * 2100: m = max(N // 8, 1)
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    t = torch.from_numpy(cloud(B, N, kind)).to(dev)
    srt, gbox = pm.spatial_sort(t)
    idx, xyz_s = pm.fps_sorted(srt, gbox, m, with_xyz=True, xyz=t if N > 12288 else None)
    ref = ops.farthest_point_sample(m, t)
    ok = bool(torch.equal(idx, ref)) and bool(torch.equal(xyz_s, torch.gather(t, 1, ref.long()[:, :, None].expand(-1, -1, 3))))
    if not ok:
        bad += 1
        print("MISMATCH", it, B, N, m, kind, flush=True)
print("cases %d mismatches %d" % (cases, bad))

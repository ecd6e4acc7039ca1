"""Randomised agreement of the pruned / batched geometry kernels with the plain ones (which are pinned to the oracle
in test_ops_gpu.py): many cloud shapes, densities and degenerate layouts, bit-exact indices."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clouds(rng, B, N, kind):
    if kind == "uniform":
        x = rng.random((B, N, 3), dtype=np.float32)
    elif kind == "blobs":
        c = rng.random((B, 6, 3), dtype=np.float32) * 50
        x = c[:, rng.integers(0, 6, N)] + rng.standard_normal((B, N, 3)).astype(np.float32) * rng.choice([0.01, 0.5, 3.0])
    elif kind == "plane":
        x = rng.random((B, N, 3), dtype=np.float32) * 30
        x[:, :, 2] *= 1e-4
    elif kind == "line":
        t = rng.random((B, N, 1), dtype=np.float32)
        x = t * np.float32([3.0, -2.0, 0.5]) + rng.standard_normal((B, N, 3)).astype(np.float32) * 1e-3
    elif kind == "grid":  # exact ties everywhere
        g = rng.integers(0, 12, (B, N, 3)).astype(np.float32)
        x = g * np.float32(0.25)
    elif kind == "dups":
        base = rng.random((B, max(N // 3, 1), 3), dtype=np.float32)
        x = base[:, rng.integers(0, base.shape[1], N)]
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(x.astype(np.float32))


KINDS = ("uniform", "blobs", "plane", "line", "grid", "dups")


@pytest.mark.parametrize("seed", range(6))
def test_fps_ordered_equals_plain_fps(dev, seed):
    from dh3d_amd import ops, pm
    rng = np.random.default_rng(100 + seed)
    for _ in range(10):
        B = int(rng.integers(1, 5))
        N = int(rng.choice([64, 100, 513, 1024, 2049, 4096, 5000, 8192, 9999, 12288]))
        m = int(rng.choice([1, 2, max(N // 8, 1), max(N // 3, 1), N]))
        if N * m > 12288 * 2048:
            m = N // 8
        kind = KINDS[int(rng.integers(0, len(KINDS)))]
        t = torch.from_numpy(_clouds(rng, B, N, kind)).to(dev)
        srt, gbox = pm.spatial_sort(t)
        idx, xyz_s = pm.fps_sorted(srt, gbox, m, with_xyz=True)
        ref = ops.farthest_point_sample(m, t)
        assert torch.equal(idx, ref), (B, N, m, kind)
        assert torch.equal(xyz_s, torch.gather(t, 1, ref.long()[:, :, None].expand(-1, -1, 3))), (B, N, m, kind)


@pytest.mark.parametrize("seed", range(6))
def test_knn_ordered_equals_bruteforce(dev, seed):
    from dh3d_amd import pm
    rng = np.random.default_rng(200 + seed)
    for _ in range(8):
        B = int(rng.integers(1, 40))
        N = int(rng.choice([65, 300, 1024, 2500, 4096, 8192]))
        if B * N > 40 * 4096:
            B = max(40 * 4096 // N, 1)
        K = int(rng.choice([1, 3, 8, 12, 16, 20]))
        kind = KINDS[int(rng.integers(0, len(KINDS)))]
        t = torch.from_numpy(_clouds(rng, B, N, kind)).to(dev)
        srt, gbox = pm.spatial_sort(t)
        nn, d = pm.knn_sorted(srt, gbox, K)   # B and N pick the split factor: 8 / 4 / 2 / 1 waves per query group
        nn0, d0 = pm.knn_xyz(t, K)
        assert torch.equal(nn, nn0) and torch.equal(d, d0), (B, N, K, kind)

"""Dev: kNN kernels on scene-like clouds (a noisy ground plane + vertical walls + clutter, normalised to [-1, 1] like the
Oxford submaps) instead of the bench's uniform cube: cells of the 16^3 grid then hold 0 or dozens of points."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")

def scene(B, N, rng):
    return bench.scene_like_clouds(B, N, int(rng.integers(1 << 30)), "cpu").numpy()

rng = np.random.default_rng(7)
for B, N in ((8, 8192), (32, 4096)):
    flat = (rng.random((B, N, 3), dtype=np.float32) * np.array([60, 60, 8], np.float32)).astype(np.float32)
    for name, pts in (("uniform cube", rng.random((B, N, 3), dtype=np.float32)), ("uniform 60x60x8", flat), ("scene-like", scene(B, N, rng))):
        p = torch.from_numpy(pts).to(dev)
        srt, gbox, cells = pm.spatial_sort_cells(p)
        ct = cells[:, :4097].cpu().numpy()
        occ = np.diff(ct, axis=1)
        t_b = bench.event_time_ms(lambda: pm.knn_xyz(p, 8), iters=5, warm=1)
        t_s = bench.event_time_ms(lambda: pm.knn_sorted(srt, gbox, 8), iters=20, warm=3)
        t_g = bench.event_time_ms(lambda: pm.knn_grid(srt, gbox, cells, 8), iters=20, warm=3)
        forced = cells.clone(); forced[:, 4106] = 0   # the cell lists whatever the sort's verdict
        t_f = bench.event_time_ms(lambda: pm.knn_grid(srt, gbox, forced, 8), iters=20, warm=3)
        a, _ = pm.knn_xyz(p, 8); g, _ = pm.knn_grid(srt, gbox, cells, 8)
        print("%2d x %d %-12s occupied cells %4.0f of 4096, points per occupied cell mean %.1f max %d | brute force %.0f us, pruned scan %.1f us, dh3d_knn_grid %.1f us (flagged clouds %d of %d; cell lists forced %.1f us; ids equal: %s)"
              % (B, N, name, (occ > 0).sum(1).mean(), occ[occ > 0].mean(), occ.max(), t_b * 1e3, t_s * 1e3, t_g * 1e3,
                 int((cells[:, 4106] != 0).sum()), B, t_f * 1e3, bool(torch.equal(a, g))))

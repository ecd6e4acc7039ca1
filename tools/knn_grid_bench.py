"""Dev: knn_grid (cell lists) against knn_sorted (pruned shared scan) at the bench shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from dh3d_amd import pm
dev = torch.device("cuda")
for B, N in ((8, 8192), (32, 4096), (4, 16384), (1, 8192)):
    pts = bench.synthetic_clouds(B, N, 2002, dev, 0)
    srt, gbox, cells = pm.spatial_sort_cells(pts)
    t_s = bench.event_time_ms(lambda: pm.knn_sorted(srt, gbox, 8), iters=20, warm=3)
    t_g = bench.event_time_ms(lambda: pm.knn_grid(srt, gbox, cells, 8), iters=20, warm=3)
    t0 = bench.event_time_ms(lambda: pm.spatial_sort(pts), iters=20, warm=3)
    t1 = bench.event_time_ms(lambda: pm.spatial_sort_cells(pts), iters=20, warm=3)
    a, _ = pm.knn_sorted(srt, gbox, 8); b, _ = pm.knn_grid(srt, gbox, cells, 8)
    print("B=%d N=%d: knn_sorted %.1f us  knn_grid %.1f us (equal ids: %s) | sort %.1f us, sort + cell table %.1f us"
          % (B, N, t_s * 1e3, t_g * 1e3, bool(torch.equal(a, b)), t0 * 1e3, t1 * 1e3))

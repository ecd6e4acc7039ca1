"""Dev: kNN kernels on the reference's demo clouds (tests/golden/demo_clouds.npz: 268.bin / 642.bin cropped like the loaders do)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")
for B, N in ((8, 8192), (4, 16384), (32, 4096)):
    p = bench.real_oxford_clouds(B, N, dev)
    srt, gbox, cells = pm.spatial_sort_cells(p)
    ct = cells[:, :4097].cpu().numpy(); occ = np.diff(ct, axis=1)
    sched = int(cells[0, 4107]); steps = [(sched >> (2 * s)) & 3 for s in range(12)] if sched else [0, 1, 2] * 4
    t_s = bench.event_time_ms(lambda: pm.knn_sorted(srt, gbox, 8), iters=20, warm=3)
    t_g = bench.event_time_ms(lambda: pm.knn_grid(srt, gbox, cells, 8), iters=20, warm=3)
    forced = cells.clone(); forced[:, 4106] = 0
    t_f = bench.event_time_ms(lambda: pm.knn_grid(srt, gbox, forced, 8), iters=20, warm=3)
    t_sort = bench.event_time_ms(lambda: pm.spatial_sort_cells(p), iters=20, warm=3)
    a, _ = pm.knn_xyz(p, 8); g, _ = pm.knn_grid(srt, gbox, forced, 8)
    print("%2d x %5d real: grid bits x/y/z %d/%d/%d, occupied cells %4.0f, pts per occupied cell mean %.1f max %d | sort %.1f us, pruned scan %.1f us, "
          "dh3d_knn_grid %.1f us (flagged %d of %d), cell lists forced %.1f us, ids equal %s"
          % (B, N, steps.count(0), steps.count(1), steps.count(2), (occ > 0).sum(1).mean(), occ[occ > 0].mean(), occ.max(), t_sort * 1e3, t_s * 1e3,
             t_g * 1e3, int((cells[:, 4106] != 0).sum()), B, t_f * 1e3, bool(torch.equal(a, g))))

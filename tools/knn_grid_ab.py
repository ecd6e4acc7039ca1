"""Dev: dh3d_knn_grid alone on the bench's clouds (cell lists) -- A/B of library variants via DH3D_HIP_LIB."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")
out = []
for B, N in ((8, 8192), (32, 4096), (32, 512), (16, 1024), (8, 2048), (2, 8192), (4, 16384)):
    p = bench.synthetic_clouds(B, N, 2002, dev, 0)[..., :3].contiguous()
    srt, gbox, cells = pm.spatial_sort_cells(p)
    t = bench.event_time_ms(lambda: pm.knn_grid(srt, gbox, cells, 8), iters=30, warm=5) * 1e3
    a, _ = pm.knn_xyz(p, 8) if N <= 8192 else (None, None)
    g, _ = pm.knn_grid(srt, gbox, cells, 8)
    out.append("%dx%d %.1f us%s" % (B, N, t, "" if a is None or torch.equal(a, g) else " MISMATCH"))
print(os.environ.get("DH3D_HIP_LIB", "product"), " | ".join(out))

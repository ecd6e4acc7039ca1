"""Dev: how many queries of the cell-list kNN take the restart path (group-box rescan) and what it costs their waves
(tools/libgrid_probe.so = knn.hip built with -DDH3D_GRID_PROBE)."""
import ctypes, os, sys
os.environ["DH3D_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgrid_probe.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from dh3d_amd import pm, _lib as L
dev = torch.device("cuda")
lib = ctypes.CDLL(os.environ["DH3D_HIP_LIB"])
for B, N in ((8, 8192), (32, 4096), (32, 512)):
    pts = bench.synthetic_clouds(B, N, 2002, dev, 0)[..., :3].contiguous()
    srt, gbox, cells = pm.spatial_sort_cells(pts)
    pm.knn_grid(srt, gbox, cells, 8); torch.cuda.synchronize()
    h = (ctypes.c_longlong * 8)(); lib.dh3d_grid_stat_read(h, 1)
    pm.knn_grid(srt, gbox, cells, 8); torch.cuda.synchronize()
    lib.dh3d_grid_stat_read(h, 1)
    a = list(h)
    print("%2d x %5d: %d of %d queries restart (%.2f %%), in %d of %d waves; wave lifetime mean %.0f max %d cycles; restart section mean %.0f max %d cycles per wave" %
          (B, N, a[0], B * N, 100.0 * a[0] / (B * N), a[1], a[5], a[3] / max(a[5], 1), a[4], a[2] / max(a[5], 1), a[6]))

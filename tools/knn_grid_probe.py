"""Dev: phase stamps of knn_grid workgroups (tools/libgrid_probe.so = the library built with -DDH3D_GRID_PROBE)."""
import ctypes, os, sys
os.environ["DH3D_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgrid_probe.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from dh3d_amd import pm, _lib as L
dev = torch.device("cuda")
for B, N in ((8, 8192), (32, 4096)):
    pts = bench.synthetic_clouds(B, N, 2002, dev, 0)
    srt, gbox, cells = pm.spatial_sort_cells(pts)
    for _ in range(3):
        pm.knn_grid(srt, gbox, cells, 8)
    torch.cuda.synchronize()
    h = (ctypes.c_longlong * 512)()
    L.lib().dh3d_grid_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.lib().dh3d_grid_probe_read(h, 512)
    a = np.array(list(h)).reshape(64, 8)[:, :7]
    ph = np.diff(a, axis=1).mean(0)
    print("%d x %d: cycles per workgroup: enumerate0 %.0f  drain0 %.0f  merge0 %.0f | enumerate1 %.0f  drain1 %.0f  merge1 %.0f | total %.0f" %
          ((B, N) + tuple(ph) + ((a[:, 6] - a[:, 0]).mean(),)))

"""Dev: per-wave lifetimes and 32-candidate steps of three_nn_pruned_kernel (tools/libnn3probe.so: pointnet2.hip built with a
-DDH3D_NN3_PROBE instrumentation that is not in the tree: see DEADENDS.md) at 8 x 8192 against 1024 samples."""
import ctypes, os, sys
os.environ["DH3D_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnn3probe.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from dh3d_amd import pm
lib = ctypes.CDLL(os.environ["DH3D_HIP_LIB"])
dev = torch.device("cuda")
B, N = 8, 8192
p = bench.synthetic_clouds(B, N, 1234, dev)[..., :3].contiguous()
srt, gbox, cells = pm.spatial_sort_cells(p)
_, xyz_s, srt_s, gbox_s, _ = pm.fps_sorted_ordered(srt, gbox, N // 8, cells=cells)
for _ in range(3): pm.three_nn_sorted(srt, gbox, srt_s, gbox_s)
torch.cuda.synchronize()
h = (ctypes.c_longlong * (8192 * 2))(); lib.dh3d_nn3_probe_read(h, 8192 * 2)
a = np.array(list(h)).reshape(8192, 2)[: B * (N // 64) * 4].reshape(-1, 4, 2)
life = a[:, :, 0].max(1); steps = a[:, :, 1].sum(1)
print("per query group: lifetime mean %.0f, percentiles 50/90/99/100 %s cycles; steps scanned mean %.1f max %d (of %d); corr %.2f" %
      (life.mean(), np.percentile(life, [50, 90, 99, 100]).astype(int), steps.mean(), steps.max(), 32, np.corrcoef(life, steps)[0, 1]))

"""A/B of the batched FPS at cfg 5's shape (4 x 16384 -> 2048) between two builds of the library (dev tool):
   DH3D_HIP_LIB=tools/libfps_old.so python tools/fps_ab.py ; python tools/fps_ab.py"""
import torch, sys
from dh3d_amd import pm
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
for B, N in ((4, 16384), (8, 8192)):
    pts = torch.rand(B, N, 3, generator=g).to(dev)
    srt, gbox = pm.spatial_sort(pts)
    f = lambda: pm.fps_sorted(srt, gbox, N // 8, with_xyz=True, xyz=pts if N > 12288 else None)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); e1.synchronize()
    print("%d x %d -> %d: %.1f us" % (B, N, N // 8, e0.elapsed_time(e1) / 10 * 1e3))

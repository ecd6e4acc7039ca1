"""Dev: A/B of two builds of csrc/fps.hip on the ordered-cloud FPS (stand-alone libraries with only fps.hip):
   python tools/fps_ab2.py tools/libfps_base.so tools/libfps_il.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from dh3d_amd import ops, pm
dev = torch.device("cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
g = torch.Generator().manual_seed(1)
libs = [(n, ctypes.CDLL(os.path.abspath(n))) for n in sys.argv[1:]]
for B, N in ((8, 8192), (32, 4096), (4, 16384)):
    pts = torch.rand(B, N, 3, generator=g).to(dev)
    srt, gbox = pm.spatial_sort(pts)
    m = N // 8
    ref = ops.farthest_point_sample(m, pts)
    out = torch.empty(B, m, dtype=torch.int32, device=dev)
    xo = torch.empty(B, m, 3, device=dev)
    for rep in range(2):
        for name, lib in libs:
            f = lambda: lib.dh3d_fps_sorted_cloud(p(srt), p(gbox), p(pts), B, N, m, p(out), p(xo), None)
            rc = f(); torch.cuda.synchronize()
            same = bool(torch.equal(out, ref))
            for _ in range(3): f()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); e1.synchronize()
            print("%2d x %5d  %-28s rc %d  %7.1f us  identical picks: %s" % (B, N, os.path.basename(name), rc, e0.elapsed_time(e1) / 20 * 1e3, same), flush=True)

"""Dev: the ordered-cloud FPS kernels side by side (mode 0 = lists + judge + streamed picks with 12 workers, 4 = with 15,
3 = lists + judge behind barriers, 2 = one candidate per wave, 1 = one pick per round): same picks as the plain op, time per launch.   python tools/fps_modes.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from dh3d_amd import ops, pm, _lib as L
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
setm = L.lib().dh3d_dev_set_fps_sorted_mode
setm.argtypes = [ctypes.c_int]
for B, N in ((8, 8192), (32, 4096), (4, 16384), (22, 4096), (1, 8192)):
    pts = torch.rand(B, N, 3, generator=g).to(dev)
    srt, gbox = pm.spatial_sort(pts)
    ref = ops.farthest_point_sample(N // 8, pts)
    for mode in (0, 4, 3, 2):
        if mode == 1 and N > 12288: continue
        setm(mode)
        f = lambda: pm.fps_sorted(srt, gbox, N // 8, with_xyz=(mode != 1), xyz=pts if N > 12288 else None)
        o = f()
        idx = o[0] if isinstance(o, tuple) else o
        same = bool(torch.equal(idx, ref))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        print("%2d x %5d -> %4d mode %d: %7.1f us  picks identical to the plain op: %s" % (B, N, N // 8, mode, e0.elapsed_time(e1) / 20 * 1e3, same), flush=True)
setm(0)

"""Dev: ordered-FPS time with the update/reduce path disabled after round 3 (= pure per-round sync chain).
Build: hipcc ... -DDH3D_FPS_PROBE=2 dh3d_amd/csrc/fps.hip -o tools/libfps_probe2.so"""
import ctypes, torch
from dh3d_amd import pm
dev = torch.device("cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
def ev(fn, iters=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters
B, N = 8, 8192
xyz = torch.rand(B, N, 3, generator=torch.Generator().manual_seed(0)).to(dev)
srt, gbox = pm.spatial_sort(xyz)
m = N // 8
out = torch.empty(B, m, dtype=torch.int32, device=dev)
for name in ("tools/libfps_probe3.so", "tools/libfps_probe2.so"):
    lib = ctypes.CDLL(name)
    for w in (4, 8, 16):
        lib.dh3d_dev_set_fps_sorted_waves(w)
        print(name, "waves", w, "%.3f ms" % ev(lambda: lib.dh3d_fps_sorted(p(srt), p(gbox), B, N, m, p(out), None)))

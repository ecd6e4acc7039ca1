"""Dev: batched ordered FPS with phases removed (results wrong, timing only).
Build: hipcc ... -DDH3D_FPS_EXP=<bits: 1 no update+reduce, 2 no judging, 4 no update, 8 no reduce; all force 1 pick/sync> dh3d_amd/csrc/fps.hip -o tools/libfps_exp<e>.so"""
import ctypes, torch
from dh3d_amd import pm, _lib
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
out = torch.empty(B, 4096, dtype=torch.int32, device=dev)
import glob
for name in [_lib.LIB_PATH] + sorted(glob.glob("tools/libfps_exp*.so")):
    lib = ctypes.CDLL(name)
    ts = [ev(lambda: lib.dh3d_fps_sorted(p(srt), p(gbox), B, N, m, p(out), None)) for m in (8, 1024, 2048)]
    print(name.split("/")[-1], "m=8 %.3f  m=1024 %.3f  m=2048 %.3f ms   -> %.3f us per pick at the margin" % (
        ts[0], ts[1], ts[2], (ts[2] - ts[1]) / 1024 * 1e3))

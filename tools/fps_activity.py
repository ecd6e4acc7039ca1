"""Dev: how many (wave, round) pairs / groups the ordered FPS actually updates.
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -Idh3d_amd/csrc -DDH3D_FPS_PROBE \
           dh3d_amd/csrc/fps.hip -o tools/libfps_probe.so"""
import ctypes, torch
from dh3d_amd import pm
dev = torch.device("cuda")
lib = ctypes.CDLL("tools/libfps_probe.so")
p = lambda t: ctypes.c_void_p(t.data_ptr())
for B, N in ((1, 8192), (1, 4096)):
    xyz = torch.rand(B, N, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    srt, gbox = pm.spatial_sort(xyz)
    m = N // 8
    out = torch.empty(B, m, dtype=torch.int32, device=dev)
    for w in (8, 16):
        lib.dh3d_dev_set_fps_sorted_waves(w)
        z = (ctypes.c_ulonglong * 2)(); lib.dh3d_fps_cnt_read(z, 1)
        lib.dh3d_fps_sorted(p(srt), p(gbox), B, N, m, p(out), None)
        torch.cuda.synchronize()
        lib.dh3d_fps_cnt_read(z, 1)
        print("N %d waves %d: rounds %d, active waves/round %.2f of %d, updated groups/round %.2f of %d" % (
            N, w, m - 1, z[0] / (m - 1), w, z[1] / (m - 1), (N + 63) // 64))

"""Dev: where a sync of the candidate-list FPS goes (cycle sums of one wave of cloud 0).
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -Idh3d_amd/csrc -ffp-contract=off \
       -fno-honor-nans -mno-amdgpu-ieee -DDH3D_FPS_PROBE=4 [-DDH3D_FPS_PROBE_WAVE=w] dh3d_amd/csrc/fps.hip \
       -o tools/libfps_list_probe_w<w>.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from dh3d_amd import pm
dev = torch.device("cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
names = ["box tests + updates", "arg-max + list + publish", "barrier 1", "judge (wave 0) / wait", "barrier 2"]
for B, N in ((8, 8192), (32, 4096)):
    xyz = torch.rand(B, N, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    srt, gbox = pm.spatial_sort(xyz)
    m = N // 8
    out = torch.empty(B, m, dtype=torch.int32, device=dev)
    for name in sys.argv[1:] or ["tools/libfps_list_probe_w0.so", "tools/libfps_list_probe_w5.so"]:
        lib = ctypes.CDLL(os.path.abspath(name))
        h0 = (ctypes.c_longlong * 32)(); h1 = (ctypes.c_longlong * 32)()
        lib.dh3d_fps_sorted(p(srt), p(gbox), B, N, m, p(out), None); torch.cuda.synchronize()
        lib.dh3d_fps_probe_read(h0)
        lib.dh3d_fps_sorted(p(srt), p(gbox), B, N, m, p(out), None); torch.cuda.synchronize()
        lib.dh3d_fps_probe_read(h1)
        d = [b - a for a, b in zip(h0, h1)]
        n = d[15]
        print("%s  %d x %d: syncs %d, picks/sync %.2f, touched %.2f of syncs, cycles/sync %.0f (total %.0f k cycles)"
              % (name, B, N, n, d[13] / n, d[14] / n, sum(d[:5]) / n, sum(d[:5]) / 1e3))
        for i, nm in enumerate(names):
            print("  %-28s %7.0f" % (nm, d[i] / n))
        if d[7]:
            print("    judge: pool load %.0f, loop %.0f (%.0f per pick), write %.0f" % (d[6] / n, d[7] / n, d[7] / d[13], d[8] / n))

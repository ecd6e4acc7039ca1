"""Dev: where the time of the streamed candidate-list FPS goes (cycle sums: judge = wave 0, one worker wave, cloud 0).
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -Idh3d_amd/csrc -ffp-contract=off \
       -fno-honor-nans -mno-amdgpu-ieee -DDH3D_FPS_PROBE=4 -DDH3D_FPS_PROBE_WAVE=5 dh3d_amd/csrc/fps.hip \
       -o tools/libfps_stream_probe_w5.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from dh3d_amd import pm
dev = torch.device("cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
for B, N in ((8, 8192), (32, 4096)):
    xyz = torch.rand(B, N, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    srt, gbox = pm.spatial_sort(xyz)
    m = N // 8
    out = torch.empty(B, m, dtype=torch.int32, device=dev)
    for name in sys.argv[1:] or ["tools/libfps_stream_probe_w5.so"]:
        lib = ctypes.CDLL(os.path.abspath(name))
        h0 = (ctypes.c_longlong * 32)(); h1 = (ctypes.c_longlong * 32)()
        lib.dh3d_fps_sorted(p(srt), p(gbox), B, N, m, p(out), None); torch.cuda.synchronize()
        lib.dh3d_fps_probe_read(h0)
        lib.dh3d_fps_sorted(p(srt), p(gbox), B, N, m, p(out), None); torch.cuda.synchronize()
        lib.dh3d_fps_probe_read(h1)
        d = [b - a for a, b in zip(h0, h1)]
        n = d[15]
        print("%s  %d x %d: rounds %d, picks/round %.2f, judge cycles/round %.0f (total %.0f k)" % (name, B, N, n, d[13] / n, sum(d[:3]) / n, sum(d[:3]) / 1e3))
        print("  judge: wait for workers %.0f, pool load %.0f, loop %.0f (%.0f per pick)" % (d[0] / n, d[1] / n, d[2] / n, d[2] / d[13]))
        print("  worker: %d passes (%.2f picks each): wait %.0f, scan + updates %.0f, %d publishes of %.0f cycles"
              % (d[19], d[23] / max(d[19], 1), d[17] / max(d[19], 1), d[18] / max(d[19], 1), d[20], d[16] / max(d[20], 1)))

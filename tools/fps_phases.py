"""Dev: where a sync of the batched ordered FPS goes (cycle sums of one wave of cloud 0).
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -Idh3d_amd/csrc -ffp-contract=off \
       -DDH3D_FPS_PROBE=4 [-DDH3D_FPS_PROBE_WAVE=w] dh3d_amd/csrc/fps.hip -o tools/libfps_probe4.so"""
import ctypes, sys, torch
from dh3d_amd import pm
dev = torch.device("cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
B, N = 8, 8192
xyz = torch.rand(B, N, 3, generator=torch.Generator().manual_seed(0)).to(dev)
srt, gbox = pm.spatial_sort(xyz)
m = N // 8
out = torch.empty(B, m, dtype=torch.int32, device=dev)
names = ["box test", "update+reduce+publish", "barrier1", "judge", "barrier2", "read picks", " judge: LDS reads", " judge: compare", " judge: atomics"]
for name in sys.argv[1:] or ["tools/libfps_probe4.so"]:
    lib = ctypes.CDLL(name)
    h0 = (ctypes.c_longlong * 32)(); h1 = (ctypes.c_longlong * 32)()
    lib.dh3d_fps_sorted(p(srt), p(gbox), B, N, m, p(out), None); torch.cuda.synchronize()
    lib.dh3d_fps_probe_read(h0)
    lib.dh3d_fps_sorted(p(srt), p(gbox), B, N, m, p(out), None); torch.cuda.synchronize()
    lib.dh3d_fps_probe_read(h1)
    d = [b - a for a, b in zip(h0, h1)]
    n = d[15]
    print(name, "syncs", n, "active", d[14], "cycles/sync %.0f" % (sum(d[:6]) / n))
    act = ["update loop", "b1/b2", "wave max", "ballot + 2nd wave max", "key + readlane", "coords lookup", "publish"]
    for i, nm in enumerate(act):
        print("   active: %-22s %7.0f  (per active sync)" % (nm, d[16 + i] / max(d[14], 1)))
    for i, nm in enumerate(names):
        if d[i] == 0: continue
        print("  %-18s %7.0f" % (nm, d[i] / n))

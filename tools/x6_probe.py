"""Cycle stamps of one workgroup of the persistent flex_conv (dev tool).
Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Idh3d_amd/csrc -DDH3D_X6_PROBE=1 \
            dh3d_amd/csrc/flex_x6.hip -o tools/libx6_probe1.so      (and =2 -> libx6_probe2.so: forced load wait)
Run:    PYTHONPATH=. python tools/x6_probe.py"""
import ctypes, sys, torch, numpy as np
from dh3d_amd import pm
dev = torch.device("cuda")
B, N, K, Din, Dout = 8, 8192, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 64
g = torch.Generator().manual_seed(1)
xyz = torch.rand(B, N, 3, generator=g).to(dev); f = torch.randn(B, N, Din, generator=g).to(dev)
nn, _ = pm.knn_xyz(xyz, K)
theta = torch.randn(3, Din, Dout, generator=g).to(dev); bias = torch.randn(Din, Dout, generator=g).to(dev)
wp3 = pm.pack_flex_weight_x3(theta, bias); out = torch.empty(B, N, Dout, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for mode in (1, 2, 3, 4):
    lib = ctypes.CDLL("tools/libx6_probe%d.so" % mode)
    for _ in range(3):
        rc = lib.dh3d_flex_conv_pm_x6_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, K, Din, Dout, None, p(out), None)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.dh3d_flex_conv_pm_x6_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, K, Din, Dout, None, p(out), None)
    e1.record(); e1.synchronize()
    print("mode %d: %.1f us per launch   (3 = producers only load, 4 = consumers skip MFMAs)" % (mode, e0.elapsed_time(e1) / 20 * 1e3))
    if mode > 2:
        continue
    h = (ctypes.c_longlong * 1024)(); lib.dh3d_x6_probe_read(h, 1024)
    a = np.array(list(h)).reshape(2, 64, 8)
    t0 = a[0, 0, 0]
    print("mode %d rc %d  (1 = as shipped, 2 = producer waits for ALL loads before each round)" % (mode, rc))
    rounds = 16 if Din == 64 else 8
    print(" producer rounds: start | wait | compute | issue | barrier")
    for r in range(8):
        s = a[0, r]
        print("  r%2d  @%6d  %5d %5d %5d %5d" % (r, s[0] - t0, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3]))
    print(" consumer tiles: start | reduce(i-1)+gemm+partials | barrier")
    for i in range(8):
        s = a[1, i]
        print("  t%2d  @%6d  %5d %5d" % (i, s[0] - t0, s[1] - s[0], s[2] - s[1]))

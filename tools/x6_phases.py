"""Cycle stamps of one workgroup of the persistent flex_conv under the elimination variants (dev tool, round 5).
Build:  for v in 0 2 4 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared -Iinclude -Idh3d_amd/csrc \
            -DDH3D_X6_PROBE=1 -DDH3D_X6_EXP=$v dh3d_amd/csrc/flex_x6.hip -o tools/libx6_ph$v.so; done
Run:    PYTHONPATH=. python tools/x6_phases.py"""
import ctypes, glob, re, sys, torch, numpy as np
from dh3d_amd import pm
dev = torch.device("cuda")
B, N, K, Din, Dout = 8, 8192, 8, 64, 64
g = torch.Generator().manual_seed(1)
xyz = torch.rand(B, N, 3, generator=g).to(dev); f = torch.randn(B, N, Din, generator=g).to(dev)
nn, _ = pm.knn_xyz(xyz, K)
theta = torch.randn(3, Din, Dout, generator=g).to(dev); bias = torch.randn(Din, Dout, generator=g).to(dev)
wp3 = pm.pack_flex_weight_x3(theta, bias); out = torch.empty(B, N, Dout, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for path in sorted(glob.glob("tools/libx6_ph*.so"), key=lambda x: int(re.search(r"ph(\d+)", x).group(1))):
    lib = ctypes.CDLL(path)
    for _ in range(5):
        rc = lib.dh3d_flex_conv_pm_x6_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, K, Din, Dout, None, p(out), None)
    torch.cuda.synchronize()
    h = (ctypes.c_longlong * 1024)(); lib.dh3d_x6_probe_read(h, 1024)
    a = np.array(list(h)).reshape(2, 64, 8)
    t0 = a[0, 0, 0]
    print("== %s (EXP bits: 2 no partial exchange, 4 producers only load, 8 consumers skip MFMAs)" % path)
    print(" producer rounds: start | wait | compute | issue | barrier     consumer tiles: start | reduce(i-1)+gemm+partials | barrier")
    for r in range(9):
        s, c = a[0, r], a[1, r]
        print("  r%2d  @%6d  %5d %5d %5d %5d        t%2d  @%6d  %5d %5d" % (r, s[0] - t0, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3],
              r, c[0] - t0, c[1] - c[0], c[2] - c[1]))

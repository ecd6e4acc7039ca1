"""Dev: per-wave arrival/departure at every barrier of one workgroup of the persistent flex_conv.
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Idh3d_amd/csrc -DDH3D_X6_PROBE=5 \
           dh3d_amd/csrc/flex_x6.hip -o tools/libx6_probe5.so"""
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
lib = ctypes.CDLL("tools/libx6_probe5.so")
lib.dh3d_flex_conv_pm_x6_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, K, Din, Dout, None, p(out), None)
torch.cuda.synchronize()
h = (ctypes.c_longlong * (12 * 16 * 2))(); lib.dh3d_x6_bar_read(h, 12 * 16 * 2)
a = np.array(list(h)).reshape(12, 16, 2)
t0 = a[:, 0, 0].min()
print("arrival time of each wave at each barrier (cycles since first arrival); * = last to arrive")
print("bar   release | " + " ".join("  w%-2d " % w for w in range(12)))
for k in range(10):
    arr = a[:, k, 0] - t0
    rel = a[:, k, 1].min() - t0
    last = arr.argmax()
    print("%2d   %7d | " % (k, rel) + " ".join(("%5d%s" % (arr[w], "*" if w == last else " ")) for w in range(12)))

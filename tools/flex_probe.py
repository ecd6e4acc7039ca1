import ctypes, torch, numpy as np
lib = ctypes.CDLL("tools/libflex_probe.so")
dev = torch.device("cuda")
B, N, K, Din, Dout = 8, 8192, 8, 64, 64
g = torch.Generator().manual_seed(1)
xyz = torch.rand(B, N, 3, generator=g).to(dev); f = torch.randn(B, N, Din, generator=g).to(dev)
nn = torch.empty(B, N, K, dtype=torch.int32, device=dev); d = torch.empty(B, N, K, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
lib.dh3d_knn_bruteforce_xyz(p(xyz), B, N, K, p(nn), p(d), None)
theta = torch.randn(3, Din, Dout, generator=g).to(dev); bias = torch.randn(Din, Dout, generator=g).to(dev)
wp = torch.empty(4 * Din, Dout, device=dev); out = torch.empty(B, N, Dout, device=dev)
lib.dh3d_pack_flex_weight(p(theta), p(bias), Din, Dout, p(wp), None)
for _ in range(3):
    lib.dh3d_flex_conv_pm_fwd(p(f), p(xyz), p(nn), p(wp), B, N, K, Din, Dout, None, p(out), None)
torch.cuda.synchronize()
h = (ctypes.c_longlong * 512)(); lib.dh3d_flex_probe_read(h, 512)
a = np.array(list(h)).reshape(64, 8)[:, :5]
ph = np.diff(a, axis=1)
print("per-WG phases (cycles): gather %.0f  barrier-wait %.0f  gemm %.0f  store %.0f   total %.0f" % tuple(list(ph.mean(0)) + [ (a[:,4]-a[:,0]).mean() ]))
print("start offsets of first 16 WGs:", ((a[:16, 0] - a[:, 0].min()) ).tolist())

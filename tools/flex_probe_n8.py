"""Dev: phase stamps of flex_conv_pm workgroups at the sampled level's shapes (tools/libflex_probe.so = the library built
with -DDH3D_FLEX_PROBE)."""
import ctypes, torch, numpy as np
lib = ctypes.CDLL("tools/libflex_probe.so")
dev = torch.device("cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
for B, N, K, Din, Dout in ((8, 1024, 8, 64, 128), (8, 1024, 8, 128, 128), (32, 512, 8, 128, 256)):
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(B, N, 3, generator=g).to(dev); f = torch.randn(B, N, Din, generator=g).to(dev)
    nn = torch.empty(B, N, K, dtype=torch.int32, device=dev); d = torch.empty(B, N, K, device=dev)
    lib.dh3d_knn_bruteforce_xyz(p(xyz), B, N, K, p(nn), p(d), None)
    theta = torch.randn(3, Din, Dout, generator=g).to(dev); bias = torch.randn(Din, Dout, generator=g).to(dev)
    wp = torch.empty(4 * Din, Dout, device=dev); out = torch.empty(B, N, Dout, device=dev)
    lib.dh3d_pack_flex_weight(p(theta), p(bias), Din, Dout, p(wp), None)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        lib.dh3d_flex_conv_pm_fwd(p(f), p(xyz), p(nn), p(wp), B, N, K, Din, Dout, None, p(out), None)
    e0.record()
    for _ in range(10):
        lib.dh3d_flex_conv_pm_fwd(p(f), p(xyz), p(nn), p(wp), B, N, K, Din, Dout, None, p(out), None)
    e1.record(); torch.cuda.synchronize()
    h = (ctypes.c_longlong * 512)(); lib.dh3d_flex_probe_read(h, 512)
    a = np.array(list(h)).reshape(64, 8)[:, :5]
    ph = np.diff(a, axis=1)
    wp3 = torch.empty(3 * 4 * Din * Dout, dtype=torch.int16, device=dev)
    lib.dh3d_pack_flex_weight_x3(p(theta), p(bias), Din, Dout, p(wp3), None)
    for _ in range(3):
        lib.dh3d_flex_conv_pm_tile_x6_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, K, Din, Dout, None, p(out), None, 0, None, None)
    e0.record()
    for _ in range(10):
        lib.dh3d_flex_conv_pm_tile_x6_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, K, Din, Dout, None, p(out), None, 0, None, None)
    e1.record(); torch.cuda.synchronize()
    h2 = (ctypes.c_longlong * 512)(); lib.dh3d_flex_tprobe_read(h2, 512)
    a2 = np.array(list(h2)).reshape(64, 8)[:, :5]
    ph2 = np.diff(a2, axis=1)
    print("   bf16x6 tile kernel: launch %.1f us; gather+split %.0f  barrier-wait %.0f  gemm %.0f  store %.0f   total %.0f" %
          ((e0.elapsed_time(e1) * 100,) + tuple(list(ph2.mean(0)) + [(a2[:, 4] - a2[:, 0]).mean()])))
    print("%d x %d  %d -> %d: launch %.1f us; per-workgroup phases (clock64 ticks): gather %.0f  barrier-wait %.0f  gemm %.0f  store %.0f   total %.0f;  start spread %.0f" %
          ((B, N, Din, Dout, e0.elapsed_time(e1) * 100) + tuple(list(ph.mean(0)) + [(a[:, 4] - a[:, 0]).mean(), a[:, 0].max() - a[:, 0].min()])))

"""Dev: phase stamps of flex_conv_tx6 workgroups (tools/libflex_probe.so = the library built with -DDH3D_FLEX_PROBE:
python tools/build_variant.py flex_probe "-DDH3D_FLEX_PROBE" flex_tx6 flex_pm) at the global step's sampled level."""
import ctypes, sys, torch, numpy as np
import os
lib = ctypes.CDLL(os.environ.get("PROBE_LIB", "tools/libflex_probe.so"))
dev = torch.device("cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for B, N, K, Din, Dout, post in ((32, 512, 8, 64, 128, 0), (32, 512, 8, 128, 128, 0), (32, 512, 8, 128, 256, 64), (1, 16384, 12, 128, 128, 0)):
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(B, N, 3, generator=g).to(dev); f = torch.randn(B, N, Din, generator=g).to(dev)
    nn = torch.empty(B, N, K, dtype=torch.int32, device=dev); d = torch.empty(B, N, K, device=dev)
    lib.dh3d_knn_bruteforce_xyz(p(xyz), B, N, K, p(nn), p(d), None)
    theta = torch.randn(3, Din, Dout, generator=g).to(dev); bias = torch.randn(Din, Dout, generator=g).to(dev)
    out = torch.empty(B, N, Dout, device=dev)
    wp3 = torch.empty(3 * 4 * Din * Dout, dtype=torch.int16, device=dev)
    lib.dh3d_pack_flex_weight_x3(p(theta), p(bias), Din, Dout, p(wp3), None)
    wpost = out2 = None
    if post:
        W = torch.randn(Dout, post, generator=g).to(dev); wpost = torch.empty(Dout * post, device=dev)
        lib.dh3d_pack_weight(p(W), Dout, post, p(wpost), None); out2 = torch.empty(B, N, post, device=dev)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    run = lambda: lib.dh3d_flex_conv_pm_tile_x6_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, K, Din, Dout, None, p(out), p(wpost), post, p(out2), None)
    for _ in range(3): run()
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    h2 = (ctypes.c_longlong * 512)(); lib.dh3d_flex_tprobe_read(h2, 512)
    a2 = np.array(list(h2)).reshape(64, 8)[:, :5].astype(np.float64) / 100.0   # s_memtime: 100 MHz -> us
    ph2 = np.diff(a2, axis=1)
    print("%2d x %4d %3d -> %3d%s: launch %.1f us; per workgroup (us): gather+split %.2f  barrier-wait %.2f  gemm %.2f  epilogue+store%s %.2f   total %.2f; first 64 workgroups start within %.2f us" %
          (B, N, Din, Dout, " + post" if post else "", e0.elapsed_time(e1) * 100, ph2[:, 0].mean(), ph2[:, 1].mean(), ph2[:, 2].mean(), " + post GEMM" if post else "", ph2[:, 3].mean(), (a2[:, 4] - a2[:, 0]).mean(), a2[:, 0].max() - a2[:, 0].min()))

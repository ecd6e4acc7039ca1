"""Dev: the unified-wave flex_conv (tools/flex_x7_experiment.hip: measured, not shipped -- DEADENDS.md) against the shipped x6 route: bit-equality and launch time.
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared -Iinclude -Idh3d_amd/csrc -DDH3D_X7_DEV \
           tools/flex_x7_experiment.hip -o tools/libx7_dev.so   (add -DDH3D_X7_PROBE -o tools/libx7_probe.so for the stage stamps)"""
import ctypes, sys, torch
from dh3d_amd import pm
dev = torch.device("cuda")
lib = ctypes.CDLL("tools/libx7_dev.so")
p = lambda t: ctypes.c_void_p(t.data_ptr())
for (B, N, Din) in ((8, 8192, 64), (8, 8192, 32), (3, 4096 + 40, 64), (1, 100, 32), (2, 8192 * 2, 64)):
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(B, N, 3, generator=g).to(dev); f = torch.randn(B, N, Din, generator=g).to(dev)
    nn, _ = pm.knn_xyz(xyz, 8)
    theta = torch.randn(3, Din, 64, generator=g).to(dev); bias = torch.randn(Din, 64, generator=g).to(dev)
    wp3 = pm.pack_flex_weight_x3(theta, bias)
    ref = pm.flex_conv_x6(f, xyz, nn, wp3, 64)
    out = torch.full((B, N, 64), float("nan"), device=dev)
    rc = lib.dh3d_flex_conv_pm_x7_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, 8, Din, 64, None, p(out), None)
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    print("B=%d N=%d Din=%d rc=%d bit-equal=%s max|diff|=%g nan=%d" % (B, N, Din, rc, same, float((out - ref).abs().nan_to_num(1e9).max()), int(out.isnan().sum())))
    if B * N >= 65536:
        for name, fn in (("x7", lambda: lib.dh3d_flex_conv_pm_x7_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, 8, Din, 64, None, p(out), None)),
                         ("x6", lambda: pm.flex_conv_x6(f, xyz, nn, wp3, 64))):
            for _ in range(5): fn()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); e1.synchronize()
            print("   %s %.1f us per launch" % (name, e0.elapsed_time(e1) / 50 * 1e3))

try:
    import numpy as np
    plib = ctypes.CDLL("tools/libx7_probe.so")
    B, N, Din = 8, 8192, 64
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(B, N, 3, generator=g).to(dev); f = torch.randn(B, N, Din, generator=g).to(dev)
    nn, _ = pm.knn_xyz(xyz, 8)
    wp3 = pm.pack_flex_weight_x3(torch.randn(3, Din, 64, generator=g).to(dev), torch.randn(Din, 64, generator=g).to(dev))
    out = torch.empty(B, N, 64, device=dev)
    for _ in range(4):
        plib.dh3d_flex_conv_pm_x7_fwd(p(f), p(xyz), p(nn), p(wp3), B, N, 8, Din, 64, None, p(out), None)
    torch.cuda.synchronize()
    h = (ctypes.c_longlong * 192)(); plib.dh3d_x7_probe_read(h, 192)
    a = np.array(list(h)).reshape(16, 12)
    print("x7 stage stamps (cycles): k-blocks 0..7 | tail | barrier   (wave 0 of block 8)")
    for it in range(8):
        d = np.diff(a[it, :11])
        print("  tile %d @%7d: " % (it, a[it, 0] - a[0, 0]) + " ".join("%5d" % v for v in d[:8]) + " | %5d | %5d   total %6d" % (d[8], d[9], a[it, 10] - a[it, 0]))
except OSError:
    pass

"""Dev micro-benchmark of the MFMA kernels at the bench shapes (PYTHONPATH=. python tools/dense_bench.py)."""
import torch
from dh3d_amd import pm
dev = torch.device("cuda")


def ev(fn, iters=30):
    for _ in range(5):
        fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator().manual_seed(0)
for B, N in ((8, 8192), (32, 4096)):
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, 8)
    print("== B=%d N=%d (%d points)" % (B, N, B * N))
    for Din, Dout in ((32, 64), (64, 64)):
        f = torch.randn(B, N, Din, generator=g).to(dev)
        wp = pm.pack_flex_weight(torch.randn(3, Din, Dout, generator=g).to(dev), torch.randn(Din, Dout, generator=g).to(dev))
        t = ev(lambda: pm.flex_conv(f, xyz, nbr, wp, Dout, act=pm.ACT_RELU))
        F = 2.0 * B * N * 4 * Din * (8 + Dout)
        print("  flex_conv %3d->%3d  %7.1f us  %5.1f TF/s" % (Din, Dout, t, F / t / 1e6))
        wp3 = pm.pack_flex_weight_x3(torch.randn(3, Din, Dout, generator=g).to(dev), torch.randn(Din, Dout, generator=g).to(dev))
        t = ev(lambda: pm.flex_conv_x6(f, xyz, nbr, wp3, Dout, act=pm.ACT_RELU))
        print("  flex_conv_x6 %3d->%3d  %7.1f us  %5.1f TF/s-equivalent" % (Din, Dout, t, F / t / 1e6))
    for C1, C2, Dout in ((64, 0, 64), (64, 0, 128), (128, 64, 128)):
        x1 = torch.randn(B * N, C1, generator=g).to(dev)
        x2 = torch.randn(B * N, C2, generator=g).to(dev) if C2 else None
        wp = pm.pack_weight(torch.randn(C1 + C2, Dout, generator=g).to(dev))
        t = ev(lambda: pm.linear(x1, wp, Dout, x2=x2, act=pm.ACT_RELU))
        F = 2.0 * B * N * (C1 + C2) * Dout
        print("  linear %3d+%3d->%3d %7.1f us  %5.1f TF/s  %6.1f GB/s" % (C1, C2, Dout, t, F / t / 1e6, 4.0 * B * N * (C1 + C2 + Dout) / t / 1e3))
        if Dout in (128, 256) and C1 % 32 == 0 and C2 % 32 == 0:
            wp3 = pm.pack_weight_x3(torch.randn(C1 + C2, Dout, generator=g).to(dev))
            t = ev(lambda: pm.linear_x6(x1, wp3, Dout, x2=x2, act=pm.ACT_RELU))
            print("  linear_x6 %3d+%3d->%3d %7.1f us  %5.1f TF/s-eq  %6.1f GB/s" % (C1, C2, Dout, t, F / t / 1e6, 4.0 * B * N * (C1 + C2 + Dout) / t / 1e3))
    x = torch.randn(B * N, 256, generator=g).to(dev)
    wp = pm.pack_weight((torch.randn(256, 1024, generator=g) / 16).to(dev))
    wfc = torch.randn(1024, generator=g).to(dev)
    t = ev(lambda: pm.mlp_head(x, wp, 1024, wfc, 0.1), iters=10)
    print("  mlp_head 256->1024->1 %7.1f us  %5.1f TF/s" % (t, 2.0 * B * N * 256 * 1024 / t / 1e6))
    xs = torch.rand(B, N // 8, 3, generator=g).to(dev)
    nbs, _ = pm.knn_xyz(xs, 8)
    for Din, Dout in ((64, 128), (128, 128), (128, 256)):
        f = torch.randn(B, N // 8, Din, generator=g).to(dev)
        wp = pm.pack_flex_weight(torch.randn(3, Din, Dout, generator=g).to(dev), torch.randn(Din, Dout, generator=g).to(dev))
        t = ev(lambda: pm.flex_conv(f, xs, nbs, wp, Dout, act=pm.ACT_RELU))
        print("  flex_conv %3d->%3d @N/8 %7.1f us  %5.1f TF/s" % (Din, Dout, t, 2.0 * B * (N // 8) * 4 * Din * (8 + Dout) / t / 1e6))
for R in (65536, 131072):
    x = torch.randn(R, 256, generator=g).to(dev)
    W = (torch.randn(256, 1024, generator=g) / 16).to(dev)
    wfc = torch.randn(1024, generator=g).to(dev)
    w6 = pm.pack_weight_x3(W); w1 = pm.pack_weight(W)
    t1 = ev(lambda: pm.mlp_head(x, w1, 1024, wfc, 0.1), iters=10)
    t6 = ev(lambda: pm.mlp_head_x6(x, w6, 1024, wfc, 0.1), iters=10)
    print("R=%d mlp_head f32-MFMA %7.1f us (%5.1f TF/s)   bf16x6 %7.1f us (%5.1f f32-equivalent TF/s)" % (
        R, t1, 2.0 * R * 256 * 1024 / t1 / 1e6, t6, 2.0 * R * 256 * 1024 / t6 / 1e6))
    import ctypes
    from dh3d_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.dh3d_dev_set_head_wc(2)
    t62 = ev(lambda: pm.mlp_head_x6(x, w6, 1024, wfc, 0.1), iters=10)
    raw.dh3d_dev_set_head_wc(4)
    print("      bf16x6 with four 64x128 waves %7.1f us" % t62)

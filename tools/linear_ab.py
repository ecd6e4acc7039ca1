"""Dev: pm.linear (exact-f32 MFMA tile kernel) against pm.linear_x6 (bf16x6 pipeline) on the shapes of the local step's
1x1 convolutions ([8*8192, 64] -> 64 / 128).   PYTHONPATH=. python tools/linear_ab.py"""
import torch
from dh3d_amd import pm
dev = torch.device("cuda")
def ev(fn, iters=30):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for R, C, D in ((65536, 64, 64), (65536, 64, 128), (65536, 128, 128), (8192, 128, 128), (131072, 64, 128)):
    x = torch.randn(R, C, device=dev); W = torch.randn(C, D, device=dev) / C ** 0.5
    sc, sh = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
    wp = pm.pack_weight(W)
    y0 = pm.linear(x, wp, D, scale=sc, shift=sh, act=pm.ACT_RELU)
    line = "R=%d %d->%d: linear %.1f us" % (R, C, D, ev(lambda: pm.linear(x, wp, D, scale=sc, shift=sh, act=pm.ACT_RELU)))
    if D in (128, 256) and C % 32 == 0:
        w3 = pm.pack_weight_x3(W)
        y1 = pm.linear_x6(x, w3, D, scale=sc, shift=sh, act=pm.ACT_RELU)
        line += "  linear_x6 %.1f us (max |diff| %.2e)" % (ev(lambda: pm.linear_x6(x, w3, D, scale=sc, shift=sh, act=pm.ACT_RELU)),
                                                          float((y0 - y1).abs().max()))
    mb = (R * C + R * D) * 4 / 1e6
    print(line + "   [%.1f MB in + out]" % mb)

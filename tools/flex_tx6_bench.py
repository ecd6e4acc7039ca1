"""Dev: the exact-f32 tile kernel (pm.flex_conv) against the bf16x6 tile kernel (pm.flex_conv_tile_x6) at the sampled
levels' shapes and cfg 5's K = 12 layer."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")
for B, N, K, Din, Dout in ((8, 1024, 8, 64, 128), (8, 1024, 8, 128, 128), (32, 512, 8, 64, 128), (32, 512, 8, 128, 128),
                           (32, 512, 8, 128, 256), (1, 16384, 12, 128, 128)):
    xyz, f, nbr, theta, bias = bench._flex_inputs(dev, B, N, K, Din, Dout)
    wp, wp3 = pm.pack_flex_weight(theta, bias), pm.pack_flex_weight_x3(theta, bias)
    fb = torch.zeros(Dout, device=dev)
    kw = dict(pre_bias=fb, scale=fb + 1, shift=fb, act=pm.ACT_RELU)
    a = pm.flex_conv(f, xyz, nbr, wp, Dout, **kw); b = pm.flex_conv_tile_x6(f, xyz, nbr, wp3, Dout, **kw)
    t1 = bench.event_time_ms(lambda: pm.flex_conv(f, xyz, nbr, wp, Dout, **kw), iters=30, warm=3)
    t2 = bench.event_time_ms(lambda: pm.flex_conv_tile_x6(f, xyz, nbr, wp3, Dout, **kw), iters=30, warm=3)
    print("%2d x %5d K=%2d %3d -> %3d: exact-f32 tile kernel %.1f us, bf16x6 tile kernel %.1f us (max rel diff %.1e)" %
          (B, N, K, Din, Dout, t1 * 1e3, t2 * 1e3, float((a - b).abs().max() / a.abs().max())))

"""Dev: launch time of the persistent flex_conv vs batch size (fixed overhead vs per-tile slope)."""
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
N, K = 8192, 8
for Din in (64, 32):
    theta = torch.randn(3, Din, 64, generator=g).to(dev); bias = torch.randn(Din, 64, generator=g).to(dev)
    wp3 = pm.pack_flex_weight_x3(theta, bias); wp = pm.pack_flex_weight(theta, bias)
    for B in (1, 2, 4, 8, 16, 32):
        xyz = torch.rand(B, N, 3, generator=g).to(dev)
        nbr, _ = pm.knn_xyz(xyz, K)
        f = torch.randn(B, N, Din, generator=g).to(dev)
        t6 = ev(lambda: pm.flex_conv_x6(f, xyz, nbr, wp3, 64, act=pm.ACT_RELU))
        t1 = ev(lambda: pm.flex_conv(f, xyz, nbr, wp, 64, act=pm.ACT_RELU))
        print("Din %d B %2d (%2d tiles/WG): x6 %6.1f us   f32 kernel %6.1f us" % (Din, B, B, t6, t1))

"""Dev: cfg 5's flex_conv 128 -> 128, K = 12 at N = 16384 -- the fused exact-f32 kernel against the two-launch form
(S materialised + bf16x6 GEMM) of the reference-signature operator at a neighbouring channel count (128 -> 132)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from dh3d_amd import pm, ops
dev = torch.device("cuda")
B, N, K, Din = 1, 16384, 12, 128
g = torch.Generator().manual_seed(5)
xyz = torch.rand(B, N, 3, generator=g).to(dev)
nbr, _ = pm.knn_xyz(xyz, K)
f = torch.randn(B, N, Din, generator=g).to(dev)
wp = pm.pack_flex_weight((torch.randn(3, Din, 128, generator=g) / Din ** 0.5).to(dev), (torch.randn(Din, 128, generator=g) / (K * Din) ** 0.5).to(dev))
print("pm.flex_conv 128 -> 128 (fused exact-f32 kernel, point-major): %.1f us" % (bench.event_time_ms(lambda: pm.flex_conv(f, xyz, nbr, wp, 128), iters=20, warm=3) * 1e3))
for Dout in (128, 132):
    theta = (torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev)
    bias = (torch.randn(Din, Dout, generator=g) / (K * Din) ** 0.5).to(dev)
    f_cf, p_cf, n_cf = f.transpose(1, 2).contiguous(), xyz.transpose(1, 2).contiguous(), nbr.transpose(1, 2).contiguous()
    with torch.no_grad():
        t = bench.event_time_ms(lambda: ops.flex_convolution(f_cf, p_cf, n_cf, theta, bias), iters=20, warm=3)
        print("ops.flex_convolution 128 -> %d (channels-first, transposes included): %.1f us" % (Dout, t * 1e3))
        tt = bench.event_time_ms(lambda: pm.transpose_last2(f_cf), iters=20, warm=3)
        print("   one transpose of the feature map: %.1f us" % (tt * 1e3))

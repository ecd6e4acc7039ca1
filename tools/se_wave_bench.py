"""Dev: se_res + pool + conv (C = 64, K = 8) alone at the bench shapes: the wave-per-32-rows kernel against the barrier
kernel (DH3D_HIP_LIB=tools/libsedev.so DH3D_SE_BARRIER=1), outputs compared bit for bit with the separate kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import event_time_ms
from dh3d_amd import pm
dev = torch.device("cuda")
g = torch.Generator().manual_seed(5)
shapes = ((8, 8192), (32, 4096), (3, 2000))
if len(sys.argv) > 2:
    shapes = ((int(sys.argv[1]), int(sys.argv[2])),)
for B, N in shapes:
    C = 64
    x = torch.randn(B, N, C, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(torch.rand(B, N, 3, generator=g).to(dev), 8)
    W1 = (torch.randn(C, C // 4, generator=g) / 8).to(dev); b1 = torch.randn(C // 4, generator=g).to(dev)
    W2 = (torch.randn(C // 4, C, generator=g) / 4).to(dev); b2 = torch.randn(C, generator=g).to(dev)
    Wc = (torch.randn(C, C, generator=g) / 8).to(dev); bc = torch.randn(C, generator=g).to(dev)
    sc = (0.5 + torch.rand(C, generator=g)).to(dev); sh = torch.randn(C, generator=g).to(dev)
    packed = pm.se_res_pack(W1, b1, W2)
    wcp = pm.pack_weight(Wc)
    y_ref = pm.se_res_pool_packed(x, nbr, *packed, b2)
    z_ref = pm.linear(y_ref, wcp, C, pre_bias=bc, scale=sc, shift=sh, act=pm.ACT_RELU)
    y, z = pm.se_res_pool_conv(x, nbr, *packed, b2, wcp, bc, sc, sh)
    t = event_time_ms(lambda: pm.se_res_pool_conv(x, nbr, *packed, b2, wcp, bc, sc, sh), iters=30)
    print("%s %d x %d: %.1f us   y bit-equal %s  z bit-equal %s  (max |dy| %.3g, |dz| %.3g)" % (
        os.environ.get("DH3D_SE_BARRIER", "wave"), B, N, t * 1e3, torch.equal(y, y_ref), torch.equal(z, z_ref),
        float((y - y_ref).abs().max()), float((z - z_ref).abs().max())), flush=True)

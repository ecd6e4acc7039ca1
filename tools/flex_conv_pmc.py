"""Launches the roofline kernel (flex_conv 64->64, B=8, N=8192, K=8) a few times; run under
rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) to get HBM-side traffic per launch."""
import torch
from dh3d_amd import pm
dev = torch.device("cuda")
B, N, K, Din, Dout = 8, 8192, 8, 64, 64
g = torch.Generator(device="cpu").manual_seed(1)
xyz = torch.rand(B, N, 3, generator=g).to(dev)
f = torch.randn(B, N, Din, generator=g).to(dev)
nbr, _ = pm.knn_xyz(xyz, K)
wp = pm.pack_flex_weight_x3((torch.randn(3, Din, Dout, generator=g) / 8).to(dev), (torch.randn(Din, Dout, generator=g) / 22).to(dev))
fb = torch.zeros(Dout, device=dev)
# a copy kernel of known size for calibrating the counters: 64 MiB read + 64 MiB write
cal = torch.empty(16 * 1024 * 1024, device=dev)
for _ in range(24):
    out = pm.flex_conv_x6(f, xyz, nbr, wp, Dout, pre_bias=fb, scale=fb + 1, shift=fb, act=pm.ACT_RELU)
    cal2 = cal.clone()
torch.cuda.synchronize()
print("done")

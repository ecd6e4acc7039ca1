"""Dev driver for tools/kernel_pmc.sh: the local step's 64 -> 128 1x1 conv at 65536 rows, a few launches."""
import torch
from dh3d_amd import pm
dev = torch.device("cuda")
R, C, D = 65536, 64, 128
x = torch.randn(R, C, device=dev); W = torch.randn(C, D, device=dev) / C ** 0.5
sc, sh = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
w3 = pm.pack_weight_x3(W)
for _ in range(4):
    pm.linear_x6(x, w3, D, scale=sc, shift=sh, act=pm.ACT_RELU)
torch.cuda.synchronize()

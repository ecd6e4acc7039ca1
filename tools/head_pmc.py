"""Launches the bf16x6 head GEMM (R=131072, 256->1024->1) a few times; run under rocprofv3 --pmc ..."""
import torch
from dh3d_amd import pm
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
R = 131072
x = torch.randn(R, 256, generator=g).to(dev)
W = (torch.randn(256, 1024, generator=g) / 16).to(dev)
wfc = torch.randn(1024, generator=g).to(dev)
w6 = pm.pack_weight_x3(W)
for _ in range(5):
    pm.mlp_head_x6(x, w6, 1024, wfc, 0.1)
torch.cuda.synchronize()
print("done")

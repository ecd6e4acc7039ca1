import torch
from dh3d_amd import pm
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
R = 131072
x1 = torch.randn(R, 64, generator=g).to(dev)
res = torch.randn(R, 128, generator=g).to(dev)
wp = pm.pack_weight(torch.randn(64, 128, generator=g).to(dev))
x3 = torch.randn(R, 128, generator=g).to(dev)
wp3 = pm.pack_weight(torch.randn(192, 128, generator=g).to(dev))
for _ in range(5):
    pm.linear(x1, wp, 128, act=pm.ACT_RELU, residual=res)
    pm.linear(x3, wp3, 128, x2=x1, act=pm.ACT_RELU)
torch.cuda.synchronize()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dh3d_amd import pm, ops
dev = torch.device("cuda")
for (B, n) in [(1, 8192), (2, 4096), (1, 4096), (2, 8192), (1, 6000)]:
    g = torch.Generator().manual_seed(n + B)
    m, C, Hd = n // 8, 256, 1024
    fine = torch.rand(B, n, 3, generator=g).to(dev)
    samp = ops.farthest_point_sample(m, fine)
    cx = torch.gather(fine, 1, samp.long()[:, :, None].expand(-1, -1, 3)).contiguous()
    d3, i3 = ops.three_nn(fine, cx)
    coarse = torch.randn(B, m, C, generator=g).to(dev)
    W = (torch.randn(C, Hd, generator=g) / C ** 0.5).to(dev)
    wfc = (torch.randn(Hd, generator=g) / Hd ** 0.5).to(dev)
    b = torch.randn(Hd, generator=g).to(dev); sc = (0.5 + torch.rand(Hd, generator=g)).to(dev); sh = torch.randn(Hd, generator=g).to(dev)
    slices = torch.cat([pm.pack_weight_x3(W[:, j:j + 256].contiguous()) for j in range(0, Hd, 256)])
    kw = dict(pre_bias=b, scale=sc, shift=sh, act=pm.ACT_RELU)
    ref = pm.interp_head(coarse, i3, d3, slices, Hd, wfc, 0.2, **kw)
    srt, _ = pm.spatial_sort(fine)
    got = pm.interp_head(coarse, i3, d3, slices, Hd, wfc, 0.2, order=srt, **kw)
    w = 1.0 / d3.double().clamp_min(1e-10); w = w / w.sum(2, keepdim=True)
    upd = (torch.gather(coarse.double(), 1, i3.long().reshape(B, -1, 1).expand(-1, -1, C)).reshape(B, n, 3, C) * w[..., None]).sum(2)
    z = torch.relu((upd @ W.double() + b.double()) * sc.double() + sh.double()) @ wfc.double() + 0.2
    ex = torch.sigmoid(z)
    e_old = (ref.double().squeeze(-1) - ex).abs(); e_new = (got.double().squeeze(-1) - ex).abs()
    bad = (e_new > 1e-5).nonzero()
    print(B, n, "old err %.2e new err %.2e bad %d" % (e_old.max(), e_new.max(), bad.shape[0]), bad[:5].tolist())

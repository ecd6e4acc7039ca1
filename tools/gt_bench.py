"""Dev: time pm.global_tail pieces at cfg 3 (32 x 4096).  Build variants:
  for k in 1 2 4 7; do (cd dh3d_amd/csrc && hipcc ... -DDH3D_GT_SKIP=$k -c dense_x6.hip -o /tmp/dx6_$k.o && hipcc -shared ... ); done
and run with DH3D_HIP_LIB=tools/libgt_skip$k.so."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import event_time_ms
from dh3d_amd import pm, ops
dev = torch.device("cuda")
B, n = 32, 4096
g = torch.Generator().manual_seed(1)
m, C, Hd, Cl, O = n // 8, 256, 1024, 64, 256
fine = torch.rand(B, n, 3, generator=g).to(dev)
samp = ops.farthest_point_sample(m, fine)
cxyz = torch.gather(fine, 1, samp.long()[:, :, None].expand(-1, -1, 3)).contiguous()
d3, i3 = ops.three_nn(fine, cxyz)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
coarse = r(B, m, C)
W = (r(C, Hd) / C ** 0.5).contiguous(); wfc = r(Hd) / Hd ** 0.5
b, sc, sh = r(Hd), (0.5 + torch.rand(Hd, generator=g)).to(dev), r(Hd)
slices = torch.cat([pm.pack_weight_x3(W[:, j:j + 256].contiguous()) for j in range(0, Hd, 256)])
wc = pm.pack_weight((r(C, Cl) / 16).contiguous()); W2 = (r(C, Cl) / 16).contiguous()
Wh, Wg = (r(C * Cl, O) / 8).contiguous(), (r(O, O) / 16).contiguous()
cs, ch = (0.5 + torch.rand(Cl, generator=g)).to(dev), 0.1 * r(Cl)
s1, h1, s2, h2 = (0.5 + torch.rand(O, generator=g)).to(dev), 0.1 * r(O), (0.5 + torch.rand(O, generator=g)).to(dev), 0.1 * r(O)
srt, _ = pm.spatial_sort(fine)
t = event_time_ms(lambda: pm.global_tail(coarse, i3, d3, srt, slices, Hd, wfc, 0.2, (b, sc, sh, pm.ACT_RELU), wc, cs, ch, W2, Wh, s1, h1, Wg, s2, h2, l2_eps=1e-8), iters=20)
t2 = event_time_ms(lambda: pm.interp_head(coarse, i3, d3, slices, Hd, wfc, 0.2, pre_bias=b, scale=sc, shift=sh, act=pm.ACT_RELU, order=srt), iters=20)
plan = pm.walk_plan(i3, d3, srt, m)
t3 = event_time_ms(lambda: pm.global_tail(coarse, i3, d3, srt, slices, Hd, wfc, 0.2, (b, sc, sh, pm.ACT_RELU), wc, cs, ch, W2, Wh, s1, h1, Wg, s2, h2, l2_eps=1e-8, plan=plan), iters=20)
t4 = event_time_ms(lambda: pm.walk_plan(i3, d3, srt, m), iters=20)
print(os.environ.get("DH3D_HIP_LIB", "default"), "global_tail %.1f us (with walk_plan: %.1f us; the plan itself %.1f)   interp_head(sorted) %.1f us" % (t * 1e3, t3 * 1e3, t4 * 1e3, t2 * 1e3))

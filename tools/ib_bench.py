"""Timing of the commuted attention head's passes (csrc/interp_train.hip) at the training shape (22 x 4096 -> 512)."""
import torch, sys
from dh3d_amd import ops, pm
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
Bt, N = 22, 4096
M = N // 8
pts = torch.rand(Bt, N, 3, generator=g).to(dev)
samp = ops.farthest_point_sample(M, pts)
cxyz = torch.gather(pts, 1, samp.long()[:, :, None].expand(-1, -1, 3)).contiguous()
d3, i3 = ops.three_nn(pts, cxyz)
order = pm.spatial_sort(pts)[0]
G = torch.randn(Bt * M, 1024, generator=g).to(dev)
H = 1024
v = [torch.randn(H, generator=g).to(dev) for _ in range(6)]
sc = (0.5 + torch.rand(H, generator=g)).to(dev)
dl = torch.randn(Bt * N, generator=g).to(dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("colstats  %.1f us" % t(lambda: pm.interp_bn_colstats(G, i3, d3, order)))
print("bwd_sums  %.1f us" % t(lambda: pm.interp_bn_bwd_sums(G, i3, d3, order, dl, v[0], v[1], sc, sc, v[2])))
print("bwd_apply %.1f us" % t(lambda: pm.interp_bn_bwd_apply(G, i3, d3, order, dl, v[0], sc, v[3], v[4], v[5])))
from dh3d_amd import _lib as L
dd = torch.clamp(d3, min=1e-10); w = ((1.0 / dd) / (1.0 / dd).sum(2, keepdim=True)).contiguous()
go = torch.randn(Bt, N, 256, generator=g).to(dev); gp = torch.empty(Bt, M, 256, device=dev)
print("interp_bwd_sorted %.1f us" % t(lambda: L.check(L.lib().dh3d_three_interpolate_bwd_sorted(Bt, N, 256, M, L.ptr(go), L.ptr(i3), L.ptr(w), L.ptr(order), L.ptr(gp), L.stream_ptr()), "x")))

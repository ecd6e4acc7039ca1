"""Dev: time pm.interp_combine at the local step's shape (8 x 8192 fine rows, 8 x 1024 coarse rows, 128 channels, l2cat
epilogue as the model uses it).  Run once per library:  DH3D_HIP_LIB=<lib> PYTHONPATH=. python tools/interp_combine_ab.py"""
import torch
from dh3d_amd import ops, pm
dev = torch.device("cuda")
B, N, M, C = 8, 8192, 1024, 128
g = torch.Generator().manual_seed(1)
pts = torch.rand(B, N, 3, generator=g).to(dev)
samp = torch.gather(pts, 1, ops.farthest_point_sample(M, pts).long()[:, :, None].expand(-1, -1, 3)).contiguous()
d3, i3 = ops.three_nn(pts, samp)
cw = torch.randn(B, M, C, generator=g).to(dev); part = torch.randn(B, N, C, generator=g).to(dev)
res = torch.randn(B, N, C, generator=g).to(dev)
sc, sh, pb = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev), torch.randn(C, device=dev)
def ev(fn, iters=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("plain    %.1f us" % ev(lambda: pm.interp_combine(cw, i3, d3, partial=part, pre_bias=pb, scale=sc, shift=sh, act=pm.ACT_RELU)))
print("residual %.1f us" % ev(lambda: pm.interp_combine(cw, i3, d3, partial=part, pre_bias=pb, scale=sc, shift=sh, act=pm.ACT_RELU, residual=res)))
print("l2cat    %.1f us" % ev(lambda: pm.interp_combine(cw, i3, d3, partial=part, pre_bias=pb, scale=sc, shift=sh, act=pm.ACT_RELU, residual=res, l2cat=(pts, 1e-8))))

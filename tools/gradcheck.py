import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from test_training_gpu import _build
from dh3d_amd.training import global_head_autograd
dev = torch.device("cuda")
m = _build(dev, seed=3)
pts = torch.rand(7, 512, 3, device=dev)
with torch.no_grad():
    geo = m._geometry(pts, None); _, local = m.compute_local(pts, _geo=geo); lv = geo.level(8, 8)
R = torch.randn(7, 256, generator=torch.Generator().manual_seed(9)).to(dev)
def loss_fn():
    d = global_head_autograd(m, pts, local, lv, bn_training=False)
    d = d * torch.rsqrt((d * d).sum(1, keepdim=True).clamp_min(1e-8))
    return (d * R).sum() + (d * d.roll(1, 0)).sum()
named = {"theta": m.global_before_assemble.flexconv_0.position_theta, "pbias": m.global_before_assemble.flexconv_0.position_bias,
         "fbias": m.global_before_assemble.flexconv_0.feature_bias, "att_W": m.globalatt.detec_conv0.W, "cluster_w": m.cluster_weights,
         "cluster_w2": m.cluster_weights2, "hidden": m.hidden1_weights, "gating": m.gating_weights}
loss = loss_fn()
grads = torch.autograd.grad(loss, list(named.values()))
gen = torch.Generator().manual_seed(0)
for (name, p), g in zip(named.items(), grads):
    d = torch.randn(p.shape, generator=gen).to(dev)
    an = (g * d).sum().item()
    for eps in (1e-2, 1e-3):
        with torch.no_grad():
            p.add_(eps * d); lp = loss_fn().item(); p.sub_(2 * eps * d); lm = loss_fn().item(); p.add_(eps * d)
        print("%-10s analytic %+.5f  numeric(eps=%g) %+.5f" % (name, an, eps, (lp - lm) / (2 * eps)))

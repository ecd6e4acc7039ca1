"""CPU (float64) check of the plan for NetVLAD's rows commuted through the up-sampling IN TRAINING (DESIGN.md section 7,
"next"): every quantity that the training step needs from the [Bt*N, 256] up-sampled rows is rewritten on the sampled
rows (M = N/8 per cloud) plus 64-wide per-point values, and compared with torch autograd on the straightforward graph
  x = interp(c);  xn = x * rsqrt(max(|x|^2, eps));  s = xn Wc;  z = BN_train(s);  a = softmax(z) * att;
  V[b] = a^T xn;  asum[b] = sum_n a.
Forward:   cw = c Wc (GEMM on sampled rows);  u = interp(cw) (64-wide walk);  s = r u,  r = rsqrt(max(|x|^2, eps));
           V[b] = A'^T c with A'[j,k] = sum_{(n,t): i_t = j} a[n,k] r[n] w_t   (scatter of 64-wide rows + GEMM).
Backward (dV, dasum given):
           E = c dV^T (GEMM);  e = interp(E);  da = r e + dasum;  datt = sum_k da p;  dz = p (da att - sum_k da att p);
           ds = BN backward of dz (needs S1 = sum dz, S2 = sum dz s_hat: one 64-wide walk);
           dcw[j] = sum_{(n,t)} w_t r ds   (64-wide scatter);   dWc = c^T dcw;
           q[n] = r^3 (sum_k ds u + sum_k a e);   dc = A' dV + dcw Wc^T - interp^T(q x)   (the last term: ONE 256-wide walk).
usage: python tools/netvlad_commute_check.py   (prints the largest relative deviations; exits non-zero above 1e-9)"""
import sys
import torch

torch.manual_seed(0)
dt = torch.float64
Bt, N, M, D, K = 3, 96, 12, 16, 8
eps_bn, eps_l2 = 1e-5, 1e-12
c = torch.randn(Bt, M, D, dtype=dt, requires_grad=True)
Wc = (torch.randn(D, K, dtype=dt) / D ** 0.5).requires_grad_()
gamma = (0.5 + torch.rand(K, dtype=dt)).requires_grad_()
beta = torch.randn(K, dtype=dt, requires_grad=True)
att = torch.rand(Bt, N, dtype=dt, requires_grad=True)
idx = torch.stack([torch.stack([torch.randperm(M)[:3] for _ in range(N)]) for _ in range(Bt)])      # [Bt,N,3]
w = torch.rand(Bt, N, 3, dtype=dt); w = w / w.sum(2, keepdim=True)
dV = torch.randn(Bt, K, D, dtype=dt)
dasum = torch.randn(Bt, K, dtype=dt)


def interp(rows):            # rows [Bt,M,C] -> [Bt,N,C]
    g = torch.gather(rows, 1, idx.reshape(Bt, N * 3, 1).expand(-1, -1, rows.shape[2])).reshape(Bt, N, 3, -1)
    return (g * w[..., None]).sum(2)


def interp_t(vals):          # adjoint: vals [Bt,N,C] -> [Bt,M,C]
    out = torch.zeros(Bt, M, vals.shape[2], dtype=dt)
    out.scatter_add_(1, idx.reshape(Bt, N * 3, 1).expand(-1, -1, vals.shape[2]),
                     (vals[:, :, None, :] * w[..., None]).reshape(Bt, N * 3, -1))
    return out


# ---- reference: autograd on the straightforward graph
x = interp(c)
xn = x * torch.rsqrt(torch.clamp((x * x).sum(2, keepdim=True), min=eps_l2))
s = xn @ Wc
mean, var = s.reshape(-1, K).mean(0), s.reshape(-1, K).var(0, unbiased=False)
z = (s - mean) * torch.rsqrt(var + eps_bn) * gamma + beta
p = torch.softmax(z, 2)
a = p * att[..., None]
V = a.transpose(1, 2) @ xn
asum = a.sum(1)
loss = (V * dV).sum() + (asum * dasum).sum()
g_c, g_Wc, g_gamma, g_beta, g_att = torch.autograd.grad(loss, [c, Wc, gamma, beta, att])

# ---- the plan: sampled rows + 64-wide per-point values
with torch.no_grad():
    cd, Wd = c.detach(), Wc.detach()
    xx = interp(cd)                                   # (inside the 256-wide walks only; never stored)
    r = torch.rsqrt(torch.clamp((xx * xx).sum(2), min=eps_l2))          # [Bt,N]
    cw = cd @ Wd                                      # GEMM on the sampled rows
    u = interp(cw)
    s2 = r[..., None] * u
    mean2, var2 = s2.reshape(-1, K).mean(0), s2.reshape(-1, K).var(0, unbiased=False)
    rstd = torch.rsqrt(var2 + eps_bn)
    shat = (s2 - mean2) * rstd
    p2 = torch.softmax(shat * gamma.detach() + beta.detach(), 2)
    a2 = p2 * att.detach()[..., None]
    Ap = interp_t(a2 * r[..., None])                  # A' [Bt,M,K]
    V2 = Ap.transpose(1, 2) @ cd
    asum2 = a2.sum(1)
    # backward
    E = cd @ dV.transpose(1, 2)                       # [Bt,M,K]
    e = interp(E)
    da = r[..., None] * e + dasum[:, None, :]
    datt2 = (da * p2).sum(2)
    dp = da * att.detach()[..., None]
    dz = p2 * (dp - (dp * p2).sum(2, keepdim=True))
    R = Bt * N
    S1, S2 = dz.reshape(-1, K).sum(0), (dz * shat).reshape(-1, K).sum(0)
    dgamma2, dbeta2 = S2, S1
    ds = gamma.detach() * rstd * (dz - S1 / R - shat * S2 / R)
    dcw = interp_t(ds * r[..., None])
    dWc2 = (cd.reshape(-1, D).t() @ dcw.reshape(-1, K))
    q = r ** 3 * ((ds * u).sum(2) + (a2 * e).sum(2))
    # (rows clamped by eps_l2 would have dr = 0: none here)
    dc2 = Ap @ dV + dcw @ Wd.t() - interp_t(q[..., None] * xx)


def rel(a_, b_):
    return float((a_ - b_).abs().max() / (b_.abs().max() + 1e-300))


errs = {"V": rel(V2, V.detach()), "asum": rel(asum2, asum.detach()), "dc": rel(dc2, g_c), "dWc": rel(dWc2, g_Wc),
        "dgamma": rel(dgamma2, g_gamma), "dbeta": rel(dbeta2, g_beta), "datt": rel(datt2, g_att)}
for k, v in errs.items():
    print("%-7s %.2e" % (k, v))
sys.exit(0 if max(errs.values()) < 1e-9 else 1)

"""Dev: can a cheap, a-priori quantity tell the pruned scan's slow query groups apart?  Per 64-query group of the demo clouds:
the probe's measured lifetime (slowest of its S waves, tools/libknn_probe.so) against box-touch counts at several inflations."""
import ctypes, sys, torch, numpy as np
sys.path.insert(0, ".")
from bench import real_oxford_clouds, synthetic_clouds
lib = ctypes.CDLL("tools/libknn_probe.so")
dev = torch.device("cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
for name in ("real", "cube"):
    B, N, S = 8, 8192, 4
    xyz = (real_oxford_clouds(B, N, dev) if name == "real" else synthetic_clouds(B, N, 1234, dev))[..., :3].contiguous()
    NG = N // 64
    srt = torch.empty(B, N, 4, device=dev); gbox = torch.empty(B, NG, 8, device=dev)
    nn = torch.empty(B, N, 8, dtype=torch.int32, device=dev); d = torch.empty(B, N, 8, device=dev)
    for _ in range(3):
        lib.dh3d_spatial_sort(p(xyz), B, N, p(srt), p(gbox), None)
        lib.dh3d_knn_sorted(p(srt), p(gbox), B, N, 8, p(nn), p(d), None)
    torch.cuda.synchronize()
    n = B * NG * S
    h = (ctypes.c_longlong * (8 * 4096))(); lib.dh3d_knn_probe_read(h, 8 * 4096)
    a = np.array(list(h)).reshape(4096, 8)[:n].reshape(B * NG, S, 8)
    life = a[:, :, 0].max(1).astype(np.float64); scanned = a[:, :, 4].sum(1)
    g = gbox.reshape(B, NG, 8); lo, hi = g[:, :, 0:3], g[:, :, 4:7]
    ext = hi - lo
    out = ["%s: lifetime mean %.0f p99 %.0f max %.0f; corr(lifetime, groups scanned) %.2f" % (name, life.mean(), np.percentile(life, 99), life.max(), np.corrcoef(life, scanned)[0, 1])]
    for infl in (0.0, 0.25, 0.5, 1.0):
        qlo, qhi = lo - infl * ext, hi + infl * ext
        gap = torch.clamp(torch.maximum(lo[:, None, :, :] - qhi[:, :, None, :], qlo[:, :, None, :] - hi[:, None, :, :]), min=0)
        touch = ((gap * gap).sum(-1) == 0).sum(-1).reshape(-1).cpu().numpy().astype(np.float64)
        top = np.argsort(-life)[: len(life) // 50]           # the slowest 2 % of the groups
        sel = np.argsort(-touch)[: len(life) // 50]          # the 2 % the proxy would pick
        out.append("inflate %.2f: corr %.2f, of the slowest 2 %% the proxy's top 2 %% catches %d / %d" % (infl, np.corrcoef(life, touch)[0, 1], len(set(top) & set(sel)), len(top)))
    # the box diagonal itself
    diag = ext.norm(dim=-1).reshape(-1).cpu().numpy()
    top = np.argsort(-life)[: len(life) // 50]; sel = np.argsort(-diag)[: len(life) // 50]
    out.append("box diagonal: corr %.2f, catches %d / %d" % (np.corrcoef(life, diag)[0, 1], len(set(top) & set(sel)), len(top)))
    print("\n   ".join(out))

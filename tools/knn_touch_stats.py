"""Dev: per 64-point Morton group, how many groups' boxes its own box touches (distance 0) -- the candidate for telling the
pruned scan's slow query groups apart (DEADENDS r6) -- on the demo clouds, the scene generator and the uniform cube."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")
for B, N in ((8, 8192), (32, 4096), (4, 16384)):
    for name, p in (("real", bench.real_oxford_clouds(B, N, dev)), ("cube", bench.synthetic_clouds(B, N, 1234, dev)[..., :3].contiguous())):
        srt, gbox = pm.spatial_sort(p)
        g = gbox.reshape(B, -1, 8)
        lo, hi = g[:, :, 0:3], g[:, :, 4:7]
        gap = torch.clamp(torch.maximum(lo[:, :, None] - hi[:, None, :], lo[:, None, :] - hi[:, :, None]), min=0)   # [B, G, G, 3]
        touch = ((gap * gap).sum(-1) == 0).sum(-1).float()    # per group
        q = np.percentile(touch.cpu().numpy(), [10, 50, 90, 99, 100])
        print("%2d x %5d %s: groups touched per group: mean %.1f, percentiles 10/50/90/99/100: %s" % (B, N, name, float(touch.mean()), q))

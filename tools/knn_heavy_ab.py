"""Dev: the pruned scan with heavy query groups split over several workgroups (csrc/knn.hip kSplitQ / kHeavyTouch): time and
bit-equality against the brute-force kernel on the demo clouds and the uniform cube.  DH3D_HIP_LIB selects the variant."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")
out = []
for B, N in ((8, 8192), (32, 4096), (4, 16384)):
    for name, p in (("real", bench.real_oxford_clouds(B, N, dev)), ("cube", bench.synthetic_clouds(B, N, 1234, dev)[..., :3].contiguous())):
        srt, gbox = pm.spatial_sort(p)
        t = bench.event_time_ms(lambda: pm.knn_sorted(srt, gbox, 8), iters=20, warm=3) * 1e3
        a, da = pm.knn_xyz(p, 8); g, dg = pm.knn_sorted(srt, gbox, 8)
        out.append("%dx%d %s %.1f us%s" % (B, N, name, t, "" if torch.equal(a, g) and torch.equal(da, dg) else " MISMATCH"))
print(os.environ.get("DH3D_HIP_LIB", "product"), " | ".join(out))

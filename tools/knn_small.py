import torch
from dh3d_amd import pm
dev = torch.device("cuda")
def ev(fn, iters=30):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B, N in ((8, 1024), (32, 512), (8, 2048), (8, 4096)):
    xyz = torch.rand(B, N, 3, device=dev)
    t1 = ev(lambda: pm.knn_xyz(xyz, 8))
    def srt():
        s, g = pm.spatial_sort(xyz); return pm.knn_sorted(s, g, 8)
    t2 = ev(srt)
    print("B %d N %d: brute %.1f us   sort+pruned %.1f us" % (B, N, t1, t2))

import sys; sys.path.insert(0, ".")
import torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")
for B, N in ((8, 8192), (32, 4096)):
    p = bench.synthetic_clouds(B, N, 1234, dev)[..., :3].contiguous()
    srt, gbox, cells = pm.spatial_sort_cells(p)
    t0 = bench.event_time_ms(lambda: pm.fps_sorted(srt, gbox, N // 8, with_xyz=True), iters=10, warm=3) * 1e3
    t1 = bench.event_time_ms(lambda: pm.fps_sorted_ordered(srt, gbox, N // 8, cells=cells), iters=10, warm=3) * 1e3
    t2 = bench.event_time_ms(lambda: pm.fps_sorted_ordered(srt, gbox, N // 8, cells=None), iters=10, warm=3) * 1e3
    print("%d x %d: fps_sorted %.1f us, ordered + cell table %.1f us, ordered without table %.1f us" % (B, N, t0, t1, t2))

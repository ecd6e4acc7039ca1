import sys; sys.path.insert(0, ".")
import torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")
for B, N in ((8, 8192), (32, 4096)):
    for name, p in (("real", bench.real_oxford_clouds(B, N, dev)), ("cube", bench.synthetic_clouds(B, N, 1234, dev)[..., :3].contiguous())):
        srt, gbox, cells = pm.spatial_sort_cells(p)
        _, xyz_s, srt_s, gbox_s, _ = pm.fps_sorted_ordered(srt, gbox, N // 8, cells=cells)
        t = bench.event_time_ms(lambda: pm.three_nn_sorted(srt, gbox, srt_s, gbox_s), iters=20, warm=3) * 1e3
        print("%2d x %5d %-5s three_nn %.1f us" % (B, N, name, t))

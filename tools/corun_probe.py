"""Dev: do the sampled set's kNN (knn_small) and three_nn co-run?  Sequential on one stream against concurrent on two."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")
B, N = 8, 8192
p = bench.synthetic_clouds(B, N, 1234, dev)[..., :3].contiguous()
srt, gbox, cells = pm.spatial_sort_cells(p)
idx, xyz_s, srt_s, gbox_s, cells_s = pm.fps_sorted_ordered(srt, gbox, N // 8, cells=cells)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def seq():
    pm.knn_xyz(xyz_s, 8); pm.three_nn_sorted(srt, gbox, srt_s, gbox_s)
def par():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): pm.knn_xyz(xyz_s, 8)
    with torch.cuda.stream(s2): pm.three_nn_sorted(srt, gbox, srt_s, gbox_s)
    cur.wait_stream(s1); cur.wait_stream(s2)
for name, fn in (("knn_small alone", lambda: pm.knn_xyz(xyz_s, 8)), ("three_nn alone", lambda: pm.three_nn_sorted(srt, gbox, srt_s, gbox_s)),
                 ("sequential", seq), ("two streams", par)):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print("%-16s %.1f us per call (graph of 10, replayed 20 times)" % (name, (time.perf_counter() - t0) / 200 * 1e6))

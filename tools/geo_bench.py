"""Dev micro-benchmark of the geometry kernels (run on the GPU box: PYTHONPATH=. python tools/geo_bench.py)."""
import ctypes
import torch
from dh3d_amd import pm, ops, _lib
dev = torch.device("cuda")
raw = ctypes.CDLL(_lib.LIB_PATH)


def ev(fn, iters=20):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for B, N in ((1, 8192), (8, 8192), (32, 4096), (8, 1024), (32, 512)):
    xyz = torch.rand(B, N, 3, device=dev)
    srt, gbox = pm.spatial_sort(xyz)
    m = max(N // 8, 1)
    res = []
    for mode in (0, 1):  # 0 = batched rounds, 1 = one pick per round
        raw.dh3d_dev_set_fps_sorted_mode(mode)
        for w in (4, 8, 16):
            raw.dh3d_dev_set_fps_sorted_waves(w)
            try:
                res.append("m%dw%d %.3f" % (mode, w, ev(lambda: pm.fps_sorted(srt, gbox, m))))
            except Exception as e:
                res.append("m%dw%d n/a" % (mode, w))
    raw.dh3d_dev_set_fps_sorted_waves(0)
    raw.dh3d_dev_set_fps_sorted_mode(0)
    res2 = []
    for w in (4, 8, 16):
        raw.dh3d_dev_set_fps_waves(w)
        try:
            res2.append("w%d %.3f" % (w, ev(lambda: ops.farthest_point_sample(m, xyz))))
        except Exception as e:
            res2.append("w%d n/a" % w)
    raw.dh3d_dev_set_fps_waves(0)
    print("   fps_bf by waves:", " ".join(res2))
    res3 = []
    for sp in (0, 2, 4, 8):  # waves per query group of the ordered kNN (0 = one-wave kernel)
        raw.dh3d_dev_set_knn_split(sp)
        res3.append("s%d %.3f" % (sp, ev(lambda: pm.knn_sorted(srt, gbox, 8))))
    raw.dh3d_dev_set_knn_split(-1)
    print("   knn_sorted by split:", " ".join(res3))
    print(B, N, "sort %.3f knn_bf %.3f knn_sorted %.3f fps_bf %.3f fps_sorted[%s] three_nn %.3f" % (
        ev(lambda: pm.spatial_sort(xyz)), ev(lambda: pm.knn_xyz(xyz, 8)), ev(lambda: pm.knn_sorted(srt, gbox, 8)),
        ev(lambda: ops.farthest_point_sample(m, xyz)), " ".join(res),
        ev(lambda: ops.three_nn(xyz, xyz[:, :m].contiguous()))))

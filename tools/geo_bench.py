"""Dev micro-benchmark of the geometry kernels.  The kNN split sweep needs the development build of the library (the
shipped one has no knobs):  make -C dh3d_amd/csrc clean && make -C dh3d_amd/csrc DEV=1 OUT=../../tools/libdh3d_dev.so
   DH3D_HIP_LIB=tools/libdh3d_dev.so PYTHONPATH=. python tools/geo_bench.py"""
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
    if hasattr(raw, "dh3d_dev_set_knn_split"):
        res3 = []
        for sp in (0, 2, 4, 8):  # waves per query group of the ordered kNN (0 = one-wave kernel)
            raw.dh3d_dev_set_knn_split(sp)
            res3.append("s%d %.3f" % (sp, ev(lambda: pm.knn_sorted(srt, gbox, 8))))
        raw.dh3d_dev_set_knn_split(-1)
        print("   knn_sorted by split:", " ".join(res3))
    samp = torch.gather(xyz, 1, pm.fps_sorted(srt, gbox, m).long()[:, :, None].expand(-1, -1, 3)).contiguous()
    srt2, gbox2 = pm.spatial_sort(samp)
    print(B, N, "sort %.3f knn_bf %.3f knn_sorted %.3f fps_bf %.3f fps_sorted %.3f three_nn %.3f three_nn_sorted %.3f" % (
        ev(lambda: pm.spatial_sort(xyz)), ev(lambda: pm.knn_xyz(xyz, 8)), ev(lambda: pm.knn_sorted(srt, gbox, 8)),
        ev(lambda: ops.farthest_point_sample(m, xyz)), ev(lambda: pm.fps_sorted(srt, gbox, m)),
        ev(lambda: ops.three_nn(xyz, samp)), ev(lambda: pm.three_nn_sorted(srt, gbox, srt2, gbox2))))

"""Dev: the pruned scan on the reference's demo clouds by waves per query group (needs a DEV build:
python tools/build_variant.py knndev "-DDH3D_DEV" knn; DH3D_HIP_LIB=tools/libknndev.so python tools/knn_real_split.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from dh3d_amd import pm, _lib
raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda")
for B, N in ((8, 8192), (32, 4096), (4, 16384)):
    for name, p in (("real", bench.real_oxford_clouds(B, N, dev)), ("cube", bench.synthetic_clouds(B, N, 1234, dev)[..., :3].contiguous())):
        srt, gbox = pm.spatial_sort(p)
        res = []
        for sp in (0, 2, 4, 8):
            raw.dh3d_dev_set_knn_split(sp)
            res.append("S=%d %.1f us" % (sp, bench.event_time_ms(lambda: pm.knn_sorted(srt, gbox, 8), iters=20, warm=3) * 1e3))
        raw.dh3d_dev_set_knn_split(-1)
        print("%2d x %5d %s: pruned scan %s" % (B, N, name, "  ".join(res)), flush=True)

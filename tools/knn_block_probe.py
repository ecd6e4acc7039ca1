import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from dh3d_amd import pm
lib = ctypes.CDLL("tools/libknn_probe_blk.so")
dev = torch.device("cuda")
for B, N in ((1, 8192), (8, 8192)):
    pts = bench.synthetic_clouds(B, N, 2002, dev, 0)
    srt, gbox, cells = pm.spatial_sort_cells(pts)
    nn = torch.empty(B, N, 8, dtype=torch.int32, device=dev); d = torch.empty(B, N, 8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.dh3d_knn_block(p(srt), p(cells), B, N, 8, p(nn), p(d), None) == 0
    torch.cuda.synchronize()
    h = (ctypes.c_ulonglong * 16)(); lib.dh3d_knn_block_probe_read(h, 16)
    print("B=%d N=%d: waves %d, waves in fallback %d, bricks scanned %d (%.1f per wave), candidates %d (%.0f per wave)"
          % (B, N, h[0], h[1], h[2], h[2] / max(h[0], 1), h[3], h[3] / max(h[0], 1)))
    w = max(h[0], 1)
    print("   per wave: total %.0f cycles | drain %.0f (%.1f drains, %.1f slots) | scan %.0f (%.1f steps) | choose %.0f"
          % (h[4] / w, h[5] / w, h[8] / w, h[9] / w, h[6] / w, h[10] / w, h[7] / w))
    break

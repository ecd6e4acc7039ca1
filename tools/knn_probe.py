"""Per-wave counters of knn_split_kernel (built with -DDH3D_KNN_PROBE, tools/gpu_knn_probe.sh) on the bench's clouds."""
import ctypes, sys, torch, numpy as np
sys.path.insert(0, ".")
from bench import synthetic_clouds, real_oxford_clouds
REAL = len(sys.argv) > 1 and sys.argv[1] == "real"
lib = ctypes.CDLL("tools/libknn_probe.so")
dev = torch.device("cuda")
for B, N in ((8, 8192), (32, 4096)):
    xyz = (real_oxford_clouds(B, N, dev) if REAL else synthetic_clouds(B, N, 1234, dev))[..., :3].contiguous()
    NG = (N + 63) // 64
    S = 4 if NG * B <= 1280 else 2
    srt = torch.empty(B, N, 4, device=dev); gbox = torch.empty(B, NG, 8, device=dev)
    nn = torch.empty(B, N, 8, dtype=torch.int32, device=dev); d = torch.empty(B, N, 8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for _ in range(3):
        lib.dh3d_spatial_sort(p(xyz), B, N, p(srt), p(gbox), None)
        lib.dh3d_knn_sorted(p(srt), p(gbox), B, N, 8, p(nn), p(d), None)
    torch.cuda.synchronize()
    n = min(4096, B * NG * S)
    h = (ctypes.c_longlong * (8 * n))()
    lib.dh3d_knn_probe_read(h, 8 * n)
    a = np.array(list(h)).reshape(n, 8)
    packed = a[:, 6].copy()   # hits + 1e3 * sparse drains + 1e6 * their slot-iterations + 1e9 * queued entries
    a[:, 6] = packed % 1000
    sparse, sparse_slots, entries = (packed // 1000) % 1000, (packed // 1000000) % 1000, packed // 1000000000
    print("B=%d N=%d S=%d, %d waves: cycles mean %.0f max %.0f | scanning %.0f | in drains %.0f | #drains %.1f "
          "#slot-iterations %.1f | #groups scanned %.1f of %d owned | 32-candidate steps with a survivor %.1f of %.1f"
          % (B, N, S, n, a[:, 0].mean(), a[:, 0].max(), a[:, 5].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(),
             a[:, 4].mean(), NG // S, a[:, 6].mean(), 2 * a[:, 4].mean()))
    print("   drains with <= 3 lanes holding more than 4 entries: %.1f of %.1f per wave, %.1f of the %.1f slot-iterations; "
          "queued entries per wave %.0f (= %.1f per lane), i.e. %.1f %% of the lane-slots of its drains"
          % (sparse.mean(), a[:, 2].mean(), sparse_slots.mean(), a[:, 3].mean(), entries.mean(), entries.mean() / 64,
             100.0 * entries.sum() / (64.0 * a[:, 3].sum())))
    g = a.reshape(-1, S, 8)
    life = g[:, :, 0].max(1)
    print("   per query group: slowest wave mean %.0f; percentiles 50/90/99/100: %s; groups scanned mean %.1f max %d"
          % (life.mean(), np.percentile(life, [50, 90, 99, 100]).astype(int), g[:, :, 4].sum(1).mean(), g[:, :, 4].sum(1).max()))
    hs, st = a[:, 7] // 1000, a[:, 7] % 1000
    print("   32-candidate steps a per-query test against the HALF-group's box would skip: %.1f of %.1f per wave (%.0f %%)"
          % (hs.mean(), st.mean(), 100.0 * hs.sum() / max(st.sum(), 1)))
    worst = np.argsort(-a[:, 0])[:5]
    for w in worst:
        print("   slow wave %d: cycles %d scan %d drain %d drains %d slots %d groups %d hits %d"
              % (w, a[w, 0], a[w, 5], a[w, 1], a[w, 2], a[w, 3], a[w, 4], a[w, 6]))
    c = np.corrcoef(a[:, 0], a[:, 4])[0, 1], np.corrcoef(a[:, 0], a[:, 3])[0, 1]
    print("   correlation of a wave's cycles with groups scanned %.2f, with slot-iterations %.2f" % c)

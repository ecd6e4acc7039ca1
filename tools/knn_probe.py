import ctypes, torch, numpy as np
lib = ctypes.CDLL("tools/libknn_probe.so")
dev = torch.device("cuda")
for N in (4096, 8192):
    xyz = torch.rand(8, N, 3, device=dev)
    NG = (N + 63) // 64
    srt = torch.empty(8, N, 4, device=dev); gbox = torch.empty(8, NG, 8, device=dev)
    nn = torch.empty(8, N, 8, dtype=torch.int32, device=dev); d = torch.empty(8, N, 8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for _ in range(2):
        lib.dh3d_spatial_sort(p(xyz), 8, N, p(srt), p(gbox), None)
        lib.dh3d_knn_sorted(p(srt), p(gbox), 8, N, 8, p(nn), p(d), None)
    torch.cuda.synchronize()
    h = (ctypes.c_longlong * (8 * NG))()
    lib.dh3d_knn_probe_read(h, 8 * NG)
    a = np.array(list(h)).reshape(NG, 8)
    print("N=%d groups=%d: per wave mean cycles %.0f (max %.0f), in drains %.0f, #drains %.1f, #slot-iterations %.1f, #groups scanned %.1f" % (
        N, NG, a[:, 0].mean(), a[:, 0].max(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(), a[:, 4].mean()))

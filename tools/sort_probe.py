"""Dev: where spatial_sort_kernel's cycles go (a -DDH3D_SORT_PROBE build of csrc/spatial.hip, s_memtime stamps of the
first and the last wave of eight clouds).  Build + run: bash tools/gpu_sort_probe.sh"""
import ctypes, sys, numpy as np, torch
lib = ctypes.CDLL("tools/libsort_probe.so")
dev = torch.device("cuda")
names = ["load xyz", "bbox: wave reduce", "bbox: block + keys", "pass0 rank", "pass0 barrier", "pass0 scan", "pass0 scatter",
         "pass0 reload", "passes 1-2", "final gather", "store + boxes"]
for B, N in ((8, 8192), (32, 4096), (8, 1024)):
    xyz = torch.rand(B, N, 3, device=dev)
    srt = torch.empty(B, N, 4, device=dev); gbox = torch.empty(B, (N + 63) // 64, 8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for _ in range(3):
        assert lib.dh3d_spatial_sort(p(xyz), B, N, p(srt), p(gbox), None) == 0
    torch.cuda.synchronize()
    h = (ctypes.c_longlong * 256)(); lib.dh3d_sort_probe_read(h, 256)
    a = np.array(list(h)).reshape(8, 2, 16)[:, :, :11].astype(np.float64)
    for w in (0, 1):
        ph = np.diff(a[:, w, :], axis=1).mean(0)
        print("B=%d N=%d wave %2d: total %.0f ticks (100 MHz: %.1f us) | " % (B, N, 15 * w, (a[:, w, 10] - a[:, w, 0]).mean(), (a[:, w, 10] - a[:, w, 0]).mean() / 100.0)
              + "  ".join("%s %.0f" % (n, v) for n, v in zip(names[:10], ph)))

"""Dev: where a workgroup of se_res_mfma_kernel<64, POOL, CONV> spends its time at the local step's shape (8 x 8192 x 64):
clock64 stamps of wave 0 of 64 workgroups (every 16th), library built with -DDH3D_SE_PROBE
    python tools/build_variant.py se_probe "-DDH3D_SE_PROBE [-D...]" dense.hip
    DH3D_HIP_LIB=tools/libse_probe.so python tools/se_res_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from dh3d_amd import pm, backbones as bb, _lib
dev = torch.device("cuda")
B, N, C = 8, 8192, 64
g = torch.Generator().manual_seed(3)
xyz = torch.rand(B, N, 3, generator=g).to(dev)
x = torch.randn(B, N, C, generator=g).to(dev)
srt, gbox, cells = pm.spatial_sort_cells(xyz)
nbr, _ = pm.knn_grid(srt, gbox, cells, 8)
se = bb.SEBlock(C).to(dev)
conv = bb.FeatureConv1d(C, C).to(dev)
for p in list(se.parameters()) + list(conv.parameters()):
    p.data.normal_(0, 0.2, generator=None)
conv.tfconv0.prepare() if hasattr(conv.tfconv0, "prepare") else None
with torch.no_grad():
    run = lambda: se.forward_on_max_pool_then_conv(x, nbr, conv)
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("se_res_mfma<64, pool, conv> 8 x 8192: %.2f us per launch" % (e0.elapsed_time(e1) / 20 * 1e3))
    h = ctypes.CDLL(os.path.abspath(_lib.LIB_PATH))
    if hasattr(h, "dh3d_se_probe_read"):
        buf = (ctypes.c_longlong * (64 * 16))()
        h.dh3d_se_probe_read(buf, 64 * 16)
        a = np.array(list(buf), dtype=np.int64).reshape(64, 16)[:, :13]
        names = ["ids -> LDS (+barrier)", "gather 8 rows x 2 passes + max", "barrier", "squeeze GEMM (2 waves)", "barrier",
                 "excite GEMM + sigmoid", "barrier", "own rows + gate + store y", "barrier", "conv GEMM", "tiles -> LDS (2 barriers)",
                 "store out2"]
        d = np.diff(a, axis=1)
        tot = a[:, 12] - a[:, 0]
        print("  clock64 = s_memtime: shader cycles (~2.1 GHz); mean over 64 workgroups; total %.0f cycles = %.2f us, start spread %.0f cycles"
              % (tot.mean(), tot.mean() / 2100.0, a[:, 0].max() - a[:, 0].min()))
        for nm, v, mx in zip(names, d.mean(0), d.max(0)):
            print("    %-36s %7.0f cycles  %6.2f us   (max %.0f)" % (nm, v, v / 2100.0, mx))

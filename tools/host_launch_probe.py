"""Dev: is the in-flight loop bound by the HOST's graph launches?  Times pipe.submit() on the host (local workload, four
deep), the K = 20 block the bench times, and a long steady-state block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
dev = torch.device("cuda")
wl = sys.argv[1] if len(sys.argv) > 1 else "local"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else (4 if wl == "local" else 3)
preset, N, B, out = {"local": ("basic_config", 8192, 8, "xyz_feat"), "global": ("global_config", 4096, 32, "globaldesc")}[wl]
model = bench.build_model(preset, dev, seed=0, num_points=N)
pts = bench.synthetic_clouds(B, N, 11, dev)
with torch.no_grad():
    pipe = model.pipeline(pts, depth=depth, outputs=(out,))
    for _ in range(200):
        pipe.submit()
    torch.cuda.synchronize()
    for K in (20, 20, 20, 400):
        hs = []
        t0 = time.perf_counter()
        for i in range(K):
            a = time.perf_counter(); pipe.submit(); hs.append(time.perf_counter() - a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs = np.array(hs) * 1e3
        print("%s depth %d, K = %3d: %.4f ms/step in all (%.0f clouds/s); host loop alone %.4f ms/step; submit() host time "
              "median %.4f, first four %s, max %.3f ms; drain after the last submit %.3f ms"
              % (wl, depth, K, (t2 - t0) / K * 1e3, B * K / (t2 - t0), (t1 - t0) / K * 1e3, np.median(hs),
                 np.round(hs[:4], 3).tolist(), hs.max(), (t2 - t1) * 1e3))

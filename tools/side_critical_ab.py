"""Dev: the global step (one at a time and two in flight) with three_nn on the main stream in front of the sampled level
(DH3D._three_nn_before_sampled_level -> True) against three_nn on the side stream behind stage 1."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from dh3d_amd.model import DH3D
dev = torch.device("cuda")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "global"]
orig = DH3D.__dict__["_three_nn_before_sampled_level"]
for name, fn in (("rule (as shipped)", orig), ("forced True", staticmethod(lambda p: True)), ("forced False", staticmethod(lambda p: False))):
    setattr(DH3D, "_three_nn_before_sampled_level", fn if name.startswith("rule") else (lambda self, p, _v=(name == "forced True"): _v))
    model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
    pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
    with torch.no_grad():
        run = model.graphed(pts, outputs=(wl["out"],))
        for _ in range(40): run()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(7):
            t0 = time.perf_counter()
            for _ in range(40): run()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 40)
        pipe = model.pipeline(pts, depth=wl["inflight"], outputs=(wl["out"],))
        for _ in range(40): pipe.submit()
        torch.cuda.synchronize()
        bestp = 1e9
        for rep in range(7):
            t0 = time.perf_counter()
            for _ in range(40): pipe.submit()
            torch.cuda.synchronize()
            bestp = min(bestp, (time.perf_counter() - t0) / 40)
    print("%-18s one at a time %.4f ms   %d in flight %.4f ms" % (name, best * 1e3, wl["inflight"], bestp * 1e3), flush=True)

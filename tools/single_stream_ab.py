"""Dev: steps in flight with each step on ONE stream (no side stream inside a step) against the two-stream step."""
import os, sys, time
os.environ["DH3D_DEBUG_KNOBS"] = "1"  # (dh3d_amd.model reads its dev knobs only under this flag)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
dev = torch.device("cuda")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "local"]
for single in (True,):
    for depth in (4, 6, 8, 12):
        model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
        model._single_stream = single
        pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
        with torch.no_grad():
            pipe = model.pipeline(pts, depth=depth, outputs=(wl["out"],))
            for _ in range(40): pipe.submit()
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(5):
                t0 = time.perf_counter()
                for _ in range(48): pipe.submit()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 48)
        print("single_stream=%s depth %d: %.4f ms per step" % (single, depth, best * 1e3), flush=True)
        del pipe, model

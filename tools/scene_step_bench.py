"""Dev: the local and global steps on scene-like clouds (tools/knn_scene_bench.py's generator) against the bench's
uniform cube: one step at a time (a replayed hipGraph) and in flight (DH3D.pipeline) -- data-dependent kernels (FPS box
pruning, kNN, three_nn) have no cliff hidden behind the uniform benchmark input."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")

def scene(B, N, rng):
    return bench.scene_like_clouds(B, N, int(rng.integers(1 << 30)), "cpu").numpy()

def timed(fn, steps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

rng = np.random.default_rng(11)
for key in ("local", "global"):
    wl = bench.WORKLOADS[key]
    model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
    for name, pts_np in (("uniform cube", rng.random((wl["B"], wl["N"], 3), dtype=np.float32)), ("scene-like", scene(wl["B"], wl["N"], rng))):
        pts = torch.from_numpy(pts_np).to(dev)
        with torch.no_grad():
            run = model.graphed(pts, outputs=(wl["out"],))
            ms = timed(lambda: run(pts))
            pipe = model.pipeline(pts, depth=wl["inflight"], outputs=(wl["out"],))
            def go():
                t = pipe.submit(pts); pipe.release(t)
            ms_f = timed(go, steps=40)
            pipe.drain()
        print("%-6s %-12s one step at a time %.4f ms (%.0f clouds/s), %d in flight %.4f ms (%.0f clouds/s)" %
              (key, name, ms, wl["B"] / ms * 1e3, wl["inflight"], ms_f, wl["B"] / ms_f * 1e3), flush=True)

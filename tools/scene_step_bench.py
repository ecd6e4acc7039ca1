"""Dev: the local and global steps on scene-like clouds (tools/knn_scene_bench.py's generator) against the bench's
uniform cube: one step at a time (a replayed hipGraph) and in flight (DH3D.pipeline) -- data-dependent kernels (FPS box
pruning, kNN, three_nn) have no cliff hidden behind the uniform benchmark input."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")

def scene(B, N, rng):
    out = np.empty((B, N, 3), np.float32)
    for b in range(B):
        n_g, n_w = int(N * 0.55), int(N * 0.35)
        g = np.stack([rng.uniform(-1, 1, n_g), rng.uniform(-1, 1, n_g), rng.normal(-0.08, 0.004, n_g)], 1)
        walls = []
        for _ in range(6):
            m = n_w // 6
            x0, y0, ang, L = rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0, np.pi), rng.uniform(0.3, 0.9)
            t = rng.uniform(0, L, m)
            walls.append(np.stack([x0 + t * np.cos(ang), y0 + t * np.sin(ang), rng.uniform(-0.08, 0.12, m)], 1) + rng.normal(0, 0.003, (m, 3)))
        w = np.concatenate(walls)
        c = rng.uniform(-1, 1, (N - n_g - len(w), 3)) * np.array([1, 1, 0.1])
        out[b] = np.clip(np.concatenate([g, w, c])[rng.permutation(N)], -1, 1)
    return out

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

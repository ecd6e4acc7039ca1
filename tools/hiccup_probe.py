"""Dev: find the one-time ~85 ms stall seen in long submit loops (r06f steady_state 12.8 k): per-submit host times."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
dev = torch.device("cuda")
wl = bench.WORKLOADS["local"]
depth = wl["inflight"]
model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
name = wl["out"]
mode = sys.argv[1] if len(sys.argv) > 1 else "zero"
with torch.no_grad():
    pipe = model.pipeline(pts, depth=depth, outputs=(name,))
    shape = tuple(pipe._runs[0].outputs[name].shape)
    host = [torch.from_numpy(np.random.default_rng(i).random((wl["B"], wl["N"], 3), dtype=np.float32)).pin_memory() for i in range(9)]
    sink = [torch.empty(shape, device=dev) for _ in range(depth)]
    torch.cuda.synchronize()
    for rep in range(3):
        ts = []
        t0 = time.perf_counter()
        for i in range(400):
            k = pipe.next_slot
            if mode == "zero":
                pipe.submit()
            else:
                pipe.submit(host[i % 9], fetch_to={name: sink[k]})
            ts.append(time.perf_counter())
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        tt = time.perf_counter() - t0
        d = np.diff(np.array([t0] + ts)) * 1e3
        big = [(int(i), round(float(v), 2)) for i, v in enumerate(d) if v > 1.0]
        print("rep %d mode %s: total %.2f ms (host %.2f), per step %.4f; submits > 1 ms: %s" % (rep, mode, tt * 1e3, th * 1e3, tt / 400 * 1e3, big[:10]), flush=True)

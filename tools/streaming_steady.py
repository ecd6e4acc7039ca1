"""Dev: the serving loop of bench.py's `value_streaming` over LONG runs -- does the host running ahead of the device hurt?
(r06f: local, 20 steps 31.3 k clouds/s, 200 steps 12.8 k.)  Usage: streaming_steady.py [local|global]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
dev = torch.device("cuda")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "local"]
depth = wl["inflight"]
model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
name = wl["out"]
with torch.no_grad():
    pipe = model.pipeline(pts, depth=depth, outputs=(name,))
    shape = tuple(pipe._runs[0].outputs[name].shape)
    host = [torch.from_numpy(np.random.default_rng(i).random((wl["B"], wl["N"], 3), dtype=np.float32)).pin_memory() for i in range(2 * depth + 1)]
    sink = [torch.empty(shape, device=dev) for _ in range(depth)]

    def run(n, mode, lag):
        """mode: 'zero' resident batches; 'h2d' host batch only; 'sink' resident + device sink; 'both'.
        lag: the host waits for the step submitted `lag` steps ago before submitting (0: never)."""
        tickets = []
        for i in range(n):
            if lag and len(tickets) == lag:
                tickets.pop(0).event.synchronize()
            k = pipe.next_slot
            kw = {}
            if mode in ("sink", "both"):
                kw["fetch_to"] = {name: sink[k]}
            tickets.append(pipe.submit(host[i % len(host)] if mode in ("h2d", "both") else None, **kw))
            if not lag:
                tickets.clear()

    def timed(n, mode, lag):
        run(2 * depth, mode, lag)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n, mode, lag)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3

    for n in (20, 200, 1000):
        for mode in ("zero", "h2d", "sink", "both"):
            for lag in (0, 2 * depth, depth):
                ms, hostms = timed(n, mode, lag)
                print("n=%4d  %-5s lag=%d   %.4f ms per step (host loop %.4f)" % (n, mode, lag, ms, hostms), flush=True)

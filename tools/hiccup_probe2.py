"""Dev: where the one-time ~85 ms stall of a serving loop lands (first event.synchronize()? a submit?)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
dev = torch.device("cuda")
wl = bench.WORKLOADS["local"]
depth = wl["inflight"]
model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
name = wl["out"]
with torch.no_grad():
    pipe = model.pipeline(pts, depth=depth, outputs=(name,))
    torch.cuda.synchronize()
    for rep in range(4):
        tickets, log = [], []
        t0 = time.perf_counter()
        for i in range(40):
            if rep >= 1 and len(tickets) == 2 * depth:
                ta = time.perf_counter()
                tk = tickets.pop(0)
                if rep == 1:
                    tk.event.synchronize()
                elif rep == 2:
                    while not tk.event.query():
                        pass
                else:
                    tk.event.synchronize()
                tb = time.perf_counter()
                if tb - ta > 1e-3:
                    log.append(("wait", i, round((tb - ta) * 1e3, 2)))
            ta = time.perf_counter()
            tickets.append(pipe.submit())
            tb = time.perf_counter()
            if tb - ta > 1e-3:
                log.append(("submit", i, round((tb - ta) * 1e3, 2)))
        ta = time.perf_counter()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        print("rep %d: total %.2f ms, final sync %.2f ms, slow calls: %s" % (rep, (tb - t0) * 1e3, (tb - ta) * 1e3, log), flush=True)

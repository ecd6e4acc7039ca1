"""Dev: A/B of a python-side environment knob (read at capture time) on the local / global step, same process.
usage: python tools/ab_env.py NAME v0 v1 ..."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda")
def step_ms(workload, steps=80):
    wl = bench.WORKLOADS[workload]
    model = bench.build_model(wl["preset"], dev, seed=0)
    pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
    with torch.no_grad():
        run = model.graphed(pts, outputs=(wl["out"],))
        for _ in range(40): run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): run()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rep in range(2):
    for v in sys.argv[2:]:
        os.environ[sys.argv[1]] = v
        print("%s=%s: local %.4f ms   global %.4f ms" % (sys.argv[1], v, step_ms("local"), step_ms("global")), flush=True)

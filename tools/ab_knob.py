"""Dev: same-process A/B of a C-side dev knob (captured into the graph at capture time) on both workloads."""
import ctypes, sys, time, torch
import bench
from dh3d_amd import _lib
raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda")
def step_ms(workload, steps=60):
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
knob = getattr(raw, sys.argv[1])
for rep in range(2):
    for v in [int(a) for a in sys.argv[2:]]:
        knob(v)
        print("%s(%d): global %.4f ms   local %.4f ms" % (sys.argv[1], v, step_ms("global"), step_ms("local")))

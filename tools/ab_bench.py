"""Dev: same-process A/B of a Python-level switch on both workloads (boxes differ by +-2 %, runs on one box by 0.1 %).
Usage on the GPU box: PYTHONPATH=. python tools/ab_bench.py"""
import time, torch
import bench
from dh3d_amd import backbones as bb
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
orig = bb.FlexConvDilate.shortcut_fusable
for rep in range(2):
    for name, fn in (("fused shortcut", orig), ("separate shortcut conv", lambda self, n: False)):
        bb.FlexConvDilate.shortcut_fusable = fn
        print("%-24s global %.4f ms   local %.4f ms" % (name, step_ms("global"), step_ms("local")))

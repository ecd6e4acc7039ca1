"""Runs the eager forward of one bench workload a few times with the side stream folded into the caller's stream, so that
every kernel of the step runs ALONE (stand-alone durations for bench.py's `roofline_step`, per-dispatch PMC counters
for tools/gpu_step_pmc.sh).  usage: python tools/step_forward.py local|global|cfg5 [iterations]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
workload = sys.argv[1] if len(sys.argv) > 1 else "local"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = bench.WORKLOADS[workload]
dev = torch.device("cuda")
model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
model._geo_stream = torch.cuda.current_stream(dev)  # one stream: no kernel of the step overlaps another
pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
with torch.no_grad():
    for _ in range(iters):
        model(pts, fetch=(wl["out"],))
        torch.cuda.synchronize()
print("done")

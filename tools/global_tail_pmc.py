"""Launches the global-descriptor forward (global_config, 32 x 4096) a few times; run under rocprofv3 --pmc ...
(tools/gpu_global_pmc.sh collects MFMA busy cycles and the L2 request / hit counters of the tail's kernels)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
dev = torch.device("cuda")
model = bench.build_model("global_config", dev, seed=0, num_points=4096)
pts = bench.synthetic_clouds(32, 4096, 3003, dev, 0)
with torch.no_grad():
    for _ in range(4):
        model(pts, fetch=("globaldesc",))
torch.cuda.synchronize()
print("done")

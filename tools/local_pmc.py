"""Launches the local-descriptor forward (basic_config, 8 x 8192) a few times; run under rocprofv3 --pmc ...
(tools/gpu_local_pmc.sh: where the cycles of the chip-wide kernels of the local step go)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
dev = torch.device("cuda")
model = bench.build_model("basic_config", dev, seed=0, num_points=8192)
pts = bench.synthetic_clouds(8, 8192, 2002, dev, 0)
with torch.no_grad():
    for _ in range(4):
        model(pts, fetch=("xyz_feat",))
torch.cuda.synchronize()
print("done")

"""Launches the ordered three_nn (8 x 8192 fine points against 8 x 1024 samples, the local step's shape) a few times; run
under rocprofv3 --pmc ... (tools/gpu_three_nn_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from dh3d_amd import pm
import bench
dev = torch.device("cuda")
xyz = bench.synthetic_clouds(8, 8192, 2002, dev, 0)
srt, gbox = pm.spatial_sort(xyz)
idx, xyz_s = pm.fps_sorted(srt, gbox, 1024, with_xyz=True)
srt2, gbox2 = pm.spatial_sort(xyz_s)
for _ in range(6):
    pm.three_nn_sorted(srt, gbox, srt2, gbox2)
torch.cuda.synchronize()
print("done")

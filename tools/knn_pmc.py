"""Launches the ordered kNN (8 x 8192, K = 8) a few times; run under rocprofv3 --pmc ... (tools/gpu_knn_pmc.sh compares
the library of the start of round 3's second half with the current one: instruction-cache requests / misses, busy cycles)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from dh3d_amd import pm
import bench
dev = torch.device("cuda")
xyz = bench.synthetic_clouds(8, 8192, 2002, dev, 0)
srt, gbox = pm.spatial_sort(xyz)
for _ in range(6):
    pm.knn_sorted(srt, gbox, 8)
torch.cuda.synchronize()
print("done")

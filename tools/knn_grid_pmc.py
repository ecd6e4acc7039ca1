import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from dh3d_amd import pm
dev = torch.device("cuda")
pts = bench.synthetic_clouds(8, 8192, 2002, dev, 0)
srt, gbox, cells = pm.spatial_sort_cells(pts)
for _ in range(3):
    pm.knn_sorted(srt, gbox, 8); pm.knn_grid(srt, gbox, cells, 8)
torch.cuda.synchronize()

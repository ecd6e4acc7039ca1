"""Launches the ordered FPS (8 x 8192 -> 1024) a few times; run under rocprofv3 --pmc <SQ counters> --kernel-trace."""
import torch
from dh3d_amd import pm
dev = torch.device("cuda")
xyz = torch.rand(8, 8192, 3, generator=torch.Generator().manual_seed(0)).to(dev)
srt, gbox = pm.spatial_sort(xyz)
for _ in range(5):
    idx = pm.fps_sorted(srt, gbox, 1024)
torch.cuda.synchronize()
print("done")

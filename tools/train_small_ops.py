"""Which torch ops launch the small kernels of an (eager) training step: torch profiler, grouped by op and by kernel.
usage: python tools/train_small_ops.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from dh3d_amd.configs import ConfigFactory
from dh3d_amd.model import DH3D
from dh3d_amd.training import QuadrupletTrainer
import bench

dev = torch.device("cuda:0")
cfg = ConfigFactory("global_config").getconfig()
cfg.batch_size, cfg.num_pos, cfg.num_neg, cfg.num_points = 1, 2, 18, 4096
model = DH3D(cfg).init_synthetic(0).to(dev).eval().prepare()
tr = QuadrupletTrainer(model, graph_step=False)
pts = bench.synthetic_clouds(22, 4096, 4, dev, 0)
for _ in range(3):
    tr.step(pts)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step(pts)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type.name == "CUDA" or not e.kernels:
        continue
    for k in e.kernels:
        rows.append((e.name, k.name[:60], k.duration))
agg = {}
for op, kn, d in rows:
    if not any(s in kn for s in ("at::native", "rocclr", "Memcpy", "Memset", "memcpy", "memset")):
        continue
    key = (op, kn)
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += d
for (op, kn), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-44s %-60s n=%3d  %7.1f us" % (op[:44], kn, n, d))

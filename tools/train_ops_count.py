"""Which torch ops does one eager training step launch? (dev tool)"""
import torch, sys, collections
sys.path.insert(0, ".")
from dh3d_amd import ConfigFactory
from dh3d_amd.model import DH3D
from dh3d_amd.training import QuadrupletTrainer
dev = torch.device("cuda")
cfg = ConfigFactory("global_config").getconfig()
cfg.batch_size, cfg.num_pos, cfg.num_neg, cfg.num_points = 1, 2, 18, 4096
m = DH3D(cfg).init_synthetic(0).to(dev).eval().prepare()
tr = QuadrupletTrainer(m, graph_step=False)
pts = torch.rand(22, 4096, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(3):
    tr.step(pts)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step(pts)
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(((e.count, e.key, getattr(e, "device_time_total", 0.0)) for e in ev if e.key.startswith("aten::") or "Backward" in e.key), reverse=True)
print("aten ops by call count (count, name, device us total)")
for c, k, t in rows[:45]:
    print("%5d  %-50s %8.1f" % (c, k[:50], t))
print("total aten calls:", sum(c for c, k, t in rows if k.startswith("aten::")))

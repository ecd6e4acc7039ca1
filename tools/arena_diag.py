"""Dev: per-step gradient differences of two identically seeded trainers (is the step-2 deviation Adam's sign-like first step?)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from test_training_gpu import _build
from dh3d_amd.training import QuadrupletTrainer
dev = torch.device("cuda")
batches = [torch.rand(7, 4096, 3, generator=torch.Generator().manual_seed(s)).to(dev) for s in (41, 42)]
for lr in (1e-3, 1e-8):
    res = []
    for rep in range(2):
        m = _build(dev, seed=51, B=1, P=2, Ng=3)
        tr = QuadrupletTrainer(m, start_lr=lr, graph_step=False)
        tr.keep_grads = True
        out = []
        for b in batches:
            loss = tr.step(b)
            out.append((loss, [g.clone() for g in tr.last_grads]))
        res.append(out)
    for step in range(2):
        worst = max(float((x - y).abs().max()) / (float(y.abs().max()) + 1e-30) for x, y in zip(res[0][step][1], res[1][step][1]))
        print("lr %g step %d: loss %.6f %.6f  worst rel grad diff %.3e" % (lr, step, res[0][step][0], res[1][step][0], worst))
        names = [n for n, p in m.named_parameters() if p.requires_grad]
        for i, (x, y) in enumerate(zip(res[0][step][1], res[1][step][1])):
            d, mx = float((x - y).abs().max()), float(y.abs().max())
            if d > 1e-3 * mx + 1e-7:
                print("   param %d %s shape %s: max|diff| %.3e max|g| %.3e" % (i, names[i] if i < len(names) else "?", tuple(x.shape), d, mx))

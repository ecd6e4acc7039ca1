import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from dh3d_amd import pm
dev = torch.device("cuda")
rng = np.random.default_rng(1)
B, N, K = 1, 4096, 8
pts = torch.from_numpy(rng.random((B, N, 3), dtype=np.float32)).to(dev)
srt, gbox, cells = pm.spatial_sort_cells(pts)
ct = cells[0].cpu().numpy()
hd = cells[0, 4100:4106].view(torch.float32).cpu().numpy()
print("header lo, scale:", hd)
s = srt[0].cpu().numpy()
lo, sc = hd[:3], hd[3:]
c64 = np.clip(((s[:, :3] - lo) * sc).astype(np.int32), 0, 63)
c16 = c64 >> 2
sp = lambda v: (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6)
cid = sp(c16[:, 0]) | (sp(c16[:, 1]) << 1) | (sp(c16[:, 2]) << 2)
print("cids sorted nondecreasing:", bool((np.diff(cid) >= 0).all()))
starts = np.searchsorted(cid, np.arange(4097))
print("cell table equal to searchsorted:", bool((starts == ct[:4097]).all()), np.nonzero(starts != ct[:4097])[0][:10], ct[:8], starts[:8])
nn_g, d_g = pm.knn_grid(srt, cells, K)
nn_b, d_b = pm.knn_xyz(pts, K)
bad = (nn_g != nn_b).any(2)[0].cpu().numpy()
print("bad queries:", bad.sum(), "of", N)
i = np.nonzero(bad)[0][:3]
for q in i:
    print(q, nn_g[0, q].cpu().numpy(), nn_b[0, q].cpu().numpy(), d_g[0, q].cpu().numpy(), d_b[0, q].cpu().numpy())

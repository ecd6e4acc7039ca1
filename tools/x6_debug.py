"""Where does the persistent flex_conv differ from the exact-f32 kernel? (dev tool)"""
import sys, torch
from dh3d_amd import pm
dev = torch.device("cuda")
Din = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(Din)
B, N, K, Dout = 8, 8192, 8, 64
xyz = torch.rand(B, N, 3, generator=g).to(dev)
nbr, _ = pm.knn_xyz(xyz, K)
f = torch.randn(B, N, Din, generator=g).to(dev)
theta = (torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev)
bias = (torch.randn(Din, Dout, generator=g) / (8 * Din) ** 0.5).to(dev)
a = pm.flex_conv(f, xyz, nbr, pm.pack_flex_weight(theta, bias), Dout).reshape(-1, Dout)
w3 = pm.pack_flex_weight_x3(theta, bias)
for rep in range(3):
    b = pm.flex_conv_x6(f, xyz, nbr, w3, Dout).reshape(-1, Dout)
    bad = ((a - b).abs() > 1e-4 * a.abs().max()).cpu()
    rows = bad.any(1).nonzero().flatten()
    print("rep", rep, "bad elements", int(bad.sum()), "bad rows", len(rows), "of", a.shape[0])
    if len(rows):
        tiles = torch.unique(rows // 32)
        T = a.shape[0] // 32; Tx = T // 8
        info = [(int(t), int(t) // Tx, (int(t) % Tx) % 31, (int(t) % Tx) // 31) for t in tiles[:12]]
        print("  tiles (id, xcd, slot, iteration):", info)
        r0 = int(rows[0]); print("  first bad row", r0, "cols", bad[r0].nonzero().flatten()[:16].tolist())
        print("  a", a[r0, :4].tolist(), "b", b[r0, :4].tolist())

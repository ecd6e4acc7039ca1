"""Elimination timing of the persistent flex_conv (dev tool): DH3D_X6_EXP bit 1 = A fragments read once per tile, 2 = no
partial tiles / reduce / store, 4 = producers only load, 8 = consumers skip the MFMAs.
Build: for m in ...: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Idh3d_amd/csrc -DDH3D_X6_EXP=$m \
           dh3d_amd/csrc/flex_x6.hip -o tools/libx6_exp$m.so"""
import ctypes, glob, re, sys, torch
from dh3d_amd import pm
dev = torch.device("cuda")
B, N, K, Din, Dout = 8, 8192, 8, 64, 64
g = torch.Generator().manual_seed(1)
xyz = torch.rand(B, N, 3, generator=g).to(dev); f = torch.randn(B, N, Din, generator=g).to(dev)
nn, _ = pm.knn_xyz(xyz, K)
theta = torch.randn(3, Din, Dout, generator=g).to(dev); bias = torch.randn(Din, Dout, generator=g).to(dev)
wp3 = pm.pack_flex_weight_x3(theta, bias); out = torch.empty(B, N, Dout, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
libs = [("shipped", "dh3d_amd/libdh3d_hip.so")] + sorted(
    (("exp %2d" % int(re.search(r"exp(\d+)", x).group(1)), x) for x in glob.glob("tools/libx6_exp*.so")), key=lambda t: int(t[0][4:]))
if len(sys.argv) < 2:  # one subprocess per library: a variant that deadlocks costs 20 s, not the whole call
    import subprocess
    for name, path in libs:
        try:
            r = subprocess.run([sys.executable, __file__, name, path], capture_output=True, text=True, timeout=20)
            print(r.stdout.strip() or r.stderr.strip()[-200:])
        except subprocess.TimeoutExpired:
            print("%-8s HUNG" % name)
    sys.exit(0)
for name, path in [(sys.argv[1], sys.argv[2])]:
    lib = ctypes.CDLL(path)
    fn = lib.dh3d_flex_conv_pm_x6_fwd
    for _ in range(3):
        fn(p(f), p(xyz), p(nn), p(wp3), B, N, K, Din, Dout, None, p(out), None)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn(p(f), p(xyz), p(nn), p(wp3), B, N, K, Din, Dout, None, p(out), None)
    e1.record(); e1.synchronize()
    print("%-8s %6.1f us per launch" % (name, e0.elapsed_time(e1) / 50 * 1e3))

"""Timings of the backward / drop-in fast-path kernels (dev helper; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import event_time_ms, _flex_inputs
from dh3d_amd import ops, pm

dev = torch.device("cuda")
B, N, K, Din, Dout = 8, 8192, 8, 64, 64
xyz, f, nbr, theta, bias = _flex_inputs(dev, B, N, K, Din, Dout)
f_cf, p_cf, nbr_cf = f.transpose(1, 2).contiguous(), xyz.transpose(1, 2).contiguous(), nbr.transpose(1, 2).contiguous()
g = torch.randn(B, N, Dout, device=dev)
g_cf = g.transpose(1, 2).contiguous()
for fast in (True, False):
    ops.FAST_PATH = fast
    with torch.no_grad():
        t = event_time_ms(lambda: ops.flex_convolution(f_cf, p_cf, nbr_cf, theta, bias), iters=10, warm=2)
        tp = event_time_ms(lambda: ops.flex_pooling(f_cf, nbr_cf), iters=10, warm=2)
    fr = f_cf.clone().requires_grad_(); th = theta.clone().requires_grad_(); bi = bias.clone().requires_grad_()
    def fb():
        out = ops.flex_convolution(fr, p_cf, nbr_cf, th, bi)
        out.backward(g_cf)
    tb = event_time_ms(fb, iters=3 if not fast else 10, warm=1)
    print("FAST_PATH=%s  ops.flex_convolution 64->64 8x8192: fwd %.3f ms, fwd+bwd %.3f ms; flex_pooling %.3f ms" % (fast, t, tb, tp))
ops.FAST_PATH = True
print("pm.flex_conv_bwd 64->64 8x8192: %.3f ms" % event_time_ms(lambda: pm.flex_conv_bwd(f, xyz, nbr, theta, bias, g), iters=10, warm=2))
# head shape of the training step: 22 clouds x 512 points, 128 -> 256
xyz2, f2, nbr2, th2, bi2 = _flex_inputs(dev, 22, 512, 8, 128, 256)
g2 = torch.randn(22, 512, 256, device=dev)
print("pm.flex_conv_bwd 128->256 22x512: %.3f ms" % event_time_ms(lambda: pm.flex_conv_bwd(f2, xyz2, nbr2, th2, bi2, g2), iters=10, warm=2))
for (Kr, M, Nn) in [(65536, 256, 64), (90112, 256, 1024), (11264, 512, 256), (22, 16384, 256), (90112, 256, 64)]:
    A = torch.randn(Kr, M, device=dev); Bm = torch.randn(Kr, Nn, device=dev)
    t = event_time_ms(lambda: pm.gemm_tn(A, Bm), iters=10, warm=2)
    t2 = event_time_ms(lambda: A.t() @ Bm, iters=10, warm=2)
    print("gemm_tn K=%d M=%d N=%d: %.3f ms = %.1f TF   (torch/rocBLAS %.3f ms)" % (Kr, M, Nn, t, 2.0 * Kr * M * Nn / t / 1e9, t2))
for (M, Kr, Nn) in [(90112, 1024, 256), (11264, 256, 512), (65536, 64, 256), (90112, 64, 256)]:
    A = torch.randn(M, Kr, device=dev); Bm = torch.randn(Kr, Nn, device=dev)
    t = event_time_ms(lambda: pm.gemm_nn(A, Bm), iters=10, warm=2)
    t2 = event_time_ms(lambda: A @ Bm, iters=10, warm=2)
    print("gemm_nn M=%d K=%d N=%d: %.3f ms = %.1f TF   (torch/rocBLAS %.3f ms)" % (M, Kr, Nn, t, 2.0 * Kr * M * Nn / t / 1e9, t2))
x = torch.randn(8, 64, 8192, device=dev)
t = event_time_ms(lambda: pm.transpose_last2(x), iters=20)
print("transpose 8x64x8192: %.3f ms = %.2f TB/s" % (t, 2 * x.numel() * 4 / t / 1e9))

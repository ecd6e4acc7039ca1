"""Time the training GEMM shapes on the bf16x6 kernel and (DH3D_GEMM_F32=1, separate process) the exact-f32 one.
usage: python tools/gemm_bench.py            (runs itself twice)"""
import os, subprocess, sys, time

SHAPES = [("nn", 11264, 256, 1024), ("nn", 11264, 1024, 256), ("tn", 11264, 256, 1024), ("nn", 90112, 256, 64),
          ("nn", 90112, 64, 256), ("tn", 90112, 256, 64), ("tn", 11264, 512, 256), ("bnn", 4096, 256, 64),
          ("bnn", 4096, 64, 256), ("btn", 4096, 64, 256)]


def main():
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from dh3d_amd import pm
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for kind, a, b, c in SHAPES:
        if kind == "nn":
            A, B = torch.randn(a, b, device=dev, generator=g), torch.randn(b, c, device=dev, generator=g)
            f, flop = (lambda: pm.gemm_nn(A, B)), 2.0 * a * b * c
        elif kind == "tn":
            A, B = torch.randn(a, b, device=dev, generator=g), torch.randn(a, c, device=dev, generator=g)
            f, flop = (lambda: pm.gemm_tn(A, B)), 2.0 * a * b * c
        elif kind == "bnn":
            A, B = torch.randn(22, a, b, device=dev, generator=g), torch.randn(22, b, c, device=dev, generator=g)
            f, flop = (lambda: pm.gemm_nn_batched(A, B)), 44.0 * a * b * c
        else:
            A, B = torch.randn(22, a, b, device=dev, generator=g), torch.randn(22, a, c, device=dev, generator=g)
            f, flop = (lambda: pm.gemm_tn_batched(A, B)), 44.0 * a * b * c
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("%-4s %6d %5d %5d  %7.1f us  %6.1f TF" % (kind, a, b, c, us, flop / us * 1e-6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        main()
    else:
        envs = ({}, {"DH3D_GEMM_KC": "16"}, {"DH3D_GEMM_KC": "32"}, {"DH3D_GEMM_F32": "1"})
        if len(sys.argv) > 1 and sys.argv[1] == "wgs":
            envs = tuple({"DH3D_GEMM_WGS": w} for w in ("256", "384", "512", "768", "1024", "1536"))
        for env in envs:
            print(env, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=dict(os.environ, **env), timeout=120)

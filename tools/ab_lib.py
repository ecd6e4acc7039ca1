"""Dev: A/B of two builds of the library on the one-step-at-a-time local / global / cfg5 step, one subprocess per library.
usage: python tools/ab_lib.py libA.so libB.so"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import time, torch, bench
    dev = torch.device("cuda")
    for workload in ("local", "global", "cfg5"):
        wl = bench.WORKLOADS[workload]
        model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
        pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
        with torch.no_grad():
            run = model.graphed(pts, outputs=(wl["out"],))
            for _ in range(40): run()
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(5):
                t0 = time.perf_counter()
                for _ in range(40): run()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 40)
        print("  %s %.4f ms" % (workload, best * 1e3), flush=True)
else:
    for rep in range(2):
        for lib in sys.argv[1:]:
            print(lib, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=dict(os.environ, DH3D_HIP_LIB=os.path.abspath(lib)), timeout=180)

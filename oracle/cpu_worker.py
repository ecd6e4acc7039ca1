"""CPU-baseline worker (TEST / MEASUREMENT INFRASTRUCTURE ONLY): runs the oracle forward on `count` synthetic clouds
in one single-threaded process and prints the seconds it took.  bench.py's cpu_baseline leg starts one of these per
host core ("one process per cloud across all host cores") -- it imports numpy and the C oracle only, never
torch and never dh3d_amd.

    python -m oracle.cpu_worker <weights.npz> <N> <count> <seed> <global:0|1>
    python -m oracle.cpu_worker cfg1 <repeats>      BASELINE config 1: knn_bruteforce + one flex_conv 32->32 on a single
                                                    cloud of N=1024, K=8 (C oracle ops, one thread): prints the best ms
"""
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"

import numpy as np  # noqa: E402


def cfg1_inputs():
    """BASELINE config 1's inputs (seed 1001, one cloud of N=1024, K=8, flex_conv 32->32), in the reference
    operators' layouts: positions [1,3,N], features [1,32,N], theta [3,32,32], bias [32,32].  tests/test_ops_gpu.py runs the
    HIP operators on exactly these arrays."""
    rng = np.random.default_rng(1001)
    pts_T = np.ascontiguousarray(rng.random((1, 1024, 3), dtype=np.float32).transpose(0, 2, 1))
    feat = rng.standard_normal((1, 32, 1024)).astype(np.float32)
    theta = rng.standard_normal((3, 32, 32)).astype(np.float32)
    bias = rng.standard_normal((32, 32)).astype(np.float32)
    return pts_T, feat, theta, bias


def cfg1(repeats):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import cpu as O
    pts_T, feat, theta, bias = cfg1_inputs()
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        nn, _ = O.knn_bruteforce(pts_T, 8)                                           # [1, N, K]
        O.flex_convolution(feat, pts_T, np.ascontiguousarray(nn.transpose(0, 2, 1)), theta, bias, center_self=True)
        best = min(best, time.perf_counter() - t0)
    print("%.6f" % (best * 1e3), flush=True)


def main(argv):
    if argv[0] == "cfg1":
        return cfg1(int(argv[1]))
    path, N, count, seed, glob = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), bool(int(argv[4]))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import model_np
    w = dict(np.load(path))
    pts = np.random.default_rng(seed).random((count, N, 3), dtype=np.float32)
    t0 = time.perf_counter()
    model_np.forward(pts, w, detection=False, extract_global=glob)
    print("%.6f" % (time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])

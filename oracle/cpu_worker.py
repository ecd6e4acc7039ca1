"""CPU-baseline worker (TEST / MEASUREMENT INFRASTRUCTURE ONLY): runs the oracle forward on `count` synthetic clouds
in one single-threaded process and prints the seconds it took.  bench.py's cpu_baseline leg starts one of these per
host core ("one process per cloud across all host cores") -- it imports numpy and the C oracle only, never
torch and never dh3d_amd.

    python -m oracle.cpu_worker <weights.npz> <N> <count> <seed> <global:0|1>
"""
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"

import numpy as np  # noqa: E402


def main(argv):
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

#!/bin/bash
# kNN A/B: tests of the ordered kNN, timings with the shipped library, then the probe build's per-wave counters
set -e
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_pm_gpu.py -q -m gpu -k "knn or sorted or geometry" -x 2>&1 | tail -3
PYTHONPATH=. timeout 300 python tools/geo_bench.py
bash tools/gpu_knn_probe.sh

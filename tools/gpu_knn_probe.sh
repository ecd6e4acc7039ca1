#!/bin/bash
# kNN probe build + run (GPU box).
set -e
cd dh3d_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -ffp-contract=off -DDH3D_KNN_PROBE"
/opt/rocm/bin/hipcc $F -shared knn.hip spatial.hip -o ../../tools/libknn_probe.so 2>&1 | head -20
cd ../..
timeout 120 python tools/knn_probe.py

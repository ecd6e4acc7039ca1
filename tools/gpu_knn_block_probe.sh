#!/bin/bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Idh3d_amd/csrc -ffp-contract=off -DDH3D_KNN_BLOCK_PROBE -shared dh3d_amd/csrc/knn.hip -o tools/libknn_probe_blk.so || exit 1
python tools/knn_block_probe.py

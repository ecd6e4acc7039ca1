#!/bin/bash
# builds the probe variant of the sort on the GPU box and prints its phase times
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Idh3d_amd/csrc -ffp-contract=off -DDH3D_SORT_PROBE -shared dh3d_amd/csrc/spatial.hip -o tools/libsort_probe.so || exit 1
python tools/sort_probe.py

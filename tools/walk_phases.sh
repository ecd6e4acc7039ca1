#!/bin/bash
# Dev: build variants of libdh3d_hip.so with parts of the global walk (interp_head_lds_kernel, csrc/dense_x6.hip)
# compiled out (-DDH3D_IH_SKIP: 1 = no per-point work in the slice loop, 2 = no row requests; -DDH3D_GT_SKIP: 1 = no
# soft assignment, 2 = no |x| pass, 4 = no MFMA scatter) into tools/libwalk_<name>.so; time them on the GPU with
#   for v in tools/libwalk_*.so; do DH3D_HIP_LIB=$v python tools/gt_bench.py | tail -1; done
# Results of the variants are WRONG by construction: timing only.
set -e
cd "$(dirname "$0")/../dh3d_amd/csrc"
OBJS="knn.o fps.o pointnet2.o spatial.o flex_generic.o flex_pm.o flex_x6.o flex_tx6.o flex_bwd.o gemm.o train.o interp_train.o netvlad_train.o dense.o dense_tail.o netvlad.o api.o"
build() {  # name, flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -I../../include -I. $2 -c dense_x6.hip -o /tmp/dx6_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/dx6_$1.o -o ../../tools/libwalk_$1.so
}
build scatter1copy "-DDH3D_IH_SCATTER_COPIES=1" &
build noatomics "-DDH3D_GT_SKIP=8" &
build noscatter "-DDH3D_GT_SKIP=4" &
wait
ls -la ../../tools/libwalk_*.so

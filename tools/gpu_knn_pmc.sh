#!/bin/bash
# knn_split_kernel<8,4> before / after the round's kNN work (8 x 8192): instruction fetch and busy counters.
#   bash tools/gpu_knn_pmc.sh -> gpurun_out/r03_d_pmc_knn.txt   (tools/libdh3d_knn_before.so: knn.hip of commit bd58f0a)
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out/r03_d_pmc_knn.txt; mkdir -p gpurun_out; : > $out
for lib in tools/libdh3d_knn_before.so dh3d_amd/libdh3d_hip.so; do
  echo "== $lib" >> $out
  for set in "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
    d=/tmp/pmc_knn_$RANDOM; rm -rf $d
    (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env DH3D_HIP_LIB="$root/$lib" PYTHONPATH="$root" python "$root/tools/knn_pmc.py" > /tmp/log_knn 2>&1)
    f=$(find $d -name "*.db" | head -1)
    [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -2 /tmp/log_knn >> $out; continue; }
    python - "$f" >> $out <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
try:
    for row in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events "
                         "where name like '%knn_split%' group by name, counter_name"):
        print("%-40s %-22s %16.0f  (n=%d)" % (row[0][:40], row[1], row[2], row[3]))
except Exception as ex:
    print("query failed:", ex)
PY
    rm -rf $d
  done
done
cat $out

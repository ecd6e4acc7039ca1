#!/bin/bash
# PMC counters of the global path's tail kernels (tools/global_tail_pmc.py launches the forward):
#   bash tools/gpu_global_pmc.sh   -> gpurun_out/r03_pmc_global_tail.txt
# Separate --pmc passes with --kernel-trace only (the guide's recipe; no sys / hip / memory tracing beside counters).
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out/r03_pmc_global_tail.txt; mkdir -p gpurun_out; : > $out
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmc_gt_$RANDOM; rm -rf $d
  (cd /tmp && timeout 240 rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/global_tail_pmc.py" > /tmp/log_gt 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -3 /tmp/log_gt >> $out; continue; }
  python - "$f" >> $out <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
try:
    for row in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events "
                         "where name like '%interp_head%' or name like '%gemm_x6%' or name like '%netvlad%' "
                         "or name like '%linear_x6_kernel<2%' or name like '%knn_split%' group by name, counter_name"):
        print("%-44s %-26s %16.0f  (n=%d)" % (row[0][:44], row[1], row[2], row[3]))
except Exception as ex:
    print("query failed:", ex)
PY
  rm -rf $d
done
cat $out

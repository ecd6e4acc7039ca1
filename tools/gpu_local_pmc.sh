#!/bin/bash
# PMC counters of the local step's chip-wide kernels (eager forward, tools/local_pmc.py) -> gpurun_out/r03_d_pmc_local.txt
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out/r03_d_pmc_local.txt; mkdir -p gpurun_out; : > $out
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmc_loc_$RANDOM; rm -rf $d
  (cd /tmp && timeout 240 rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/local_pmc.py" > /tmp/log_loc 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -2 /tmp/log_loc >> $out; continue; }
  python - "$f" >> $out <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
try:
    for row in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events "
                         "where name like '%se_res_mfma%' or name like '%interp_combine%' or name like '%flex_conv_pm_kernel%' "
                         "or name like '%linear_x6_kernel%' or name like '%linear_pm_kernel%' or name like '%conv_pointset_pm%' "
                         "group by name, counter_name"):
        print("%-60s %-26s %14.0f  (n=%d)" % (row[0].replace("(anonymous namespace)::", "").replace("void ", "")[:60], row[1], row[2], row[3]))
except Exception as ex:
    print("query failed:", ex)
PY
  rm -rf $d
done
cat $out

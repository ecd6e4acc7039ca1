#!/bin/bash
# three_nn_sorted_kernel (8 x 8192 against 8 x 1024): where do its cycles go?  -> gpurun_out/r03_d_pmc_three_nn.txt
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out/r03_d_pmc_three_nn.txt; mkdir -p gpurun_out; : > $out
for set in "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=/tmp/pmc_3nn_$RANDOM; rm -rf $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/three_nn_pmc.py" > /tmp/log_3nn 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -2 /tmp/log_3nn >> $out; continue; }
  python - "$f" >> $out <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
try:
    for row in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events "
                         "where name like '%three_nn_sorted%' group by name, counter_name"):
        print("%-36s %-24s %16.0f  (n=%d)" % (row[0][:36], row[1], row[2], row[3]))
except Exception as ex:
    print("query failed:", ex)
PY
  rm -rf $d
done
cat $out

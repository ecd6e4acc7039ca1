#!/bin/bash
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out/${1:-r04}_pmc_knn_grid.txt; : > $out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pmc_kg_$RANDOM; rm -rf $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/knn_grid_pmc.py" > /tmp/log_kg 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -2 /tmp/log_kg >> $out; continue; }
  python - "$f" >> $out <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for row in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%knn_grid%' or name like '%knn_split%' group by name, counter_name"):
    print("%-40s %-30s %16.0f (n=%d)" % (row[0].replace("(anonymous namespace)::", "")[:40], row[1], row[2], row[3]))
PY
  rm -rf $d
done
cat $out

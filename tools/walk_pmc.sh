#!/bin/bash
# SQ counters of the sorted-walk kernels (tools/ib_bench.py launches them): bash tools/walk_pmc.sh
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out/walk_pmc.txt; : > $out
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  d=/tmp/pmc_walk_$RANDOM; rm -rf $d
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/ib_bench.py" > /tmp/log_walk 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -3 /tmp/log_walk >> $out; continue; }
  python - "$f" >> $out <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
try:
    for row in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events "
                         "where name like '%interp_bn%' group by name, counter_name"):
        print("%-36s %-26s %16.0f  (n=%d)" % (row[0][:36], row[1], row[2], row[3]))
except Exception as ex:
    print("query failed:", ex)
PY
done
cat $out

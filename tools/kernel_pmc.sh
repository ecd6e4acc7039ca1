#!/bin/bash
# Dev: PMC counters of the kernels matching a name pattern, launched by a python driver:
#   bash tools/kernel_pmc.sh <tag> <sql-like pattern> <driver.py> [driver args]   -> gpurun_out/<tag>_pmc.txt
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
tag=$1; pat=$2; shift 2
out=gpurun_out/${tag}_pmc.txt; mkdir -p gpurun_out; : > $out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmc_k_$RANDOM; rm -rf $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/$1" "${@:2}" > /tmp/log_k 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -2 /tmp/log_k >> $out; continue; }
  python - "$f" "$pat" >> $out <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for row in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like ? group by name, counter_name", (sys.argv[2],)):
    print("%-48s %-30s %16.0f (n=%d)" % (row[0].replace("(anonymous namespace)::", "")[:48], row[1], row[2], row[3]))
PY
  rm -rf $d
done
cat $out

#!/bin/bash
# SQ counters of the bf16x6 training GEMM (tools/gemm_bench.py launches the ten training shapes): bash tools/gemm_pmc.sh
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out/gemm_pmc.txt; : > $out
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  d=/tmp/pmc_gemm_$RANDOM; rm -rf $d
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/gemm_bench.py" run > /tmp/log_gemm 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -3 /tmp/log_gemm >> $out; continue; }
  python - "$f" >> $out <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
try:
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
    ev = [t for t in tabs if t.startswith("pmc_events")] or ["pmc_events"]
    for row in c.execute("select name, counter_name, avg(counter_value), count(*) from %s "
                         "where name like '%%gemm_x6%%' group by name, counter_name" % ev[0]):
        print("%-44s %-26s %16.0f  (n=%d)" % (row[0][:44], row[1], row[2], row[3]))
except Exception as ex:
    print("query failed:", ex)
PY
done
cat $out

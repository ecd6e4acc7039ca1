#!/bin/bash
# HBM traffic of EVERY kernel of one step (tools/step_forward.py, eager, one stream): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes + --kernel-trace.  bash tools/gpu_step_traffic.sh <tag> <workload>  -> gpurun_out/<tag>_traffic_<workload>.txt
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
tag=${1:-rXX}; wl=${2:-local}
out=gpurun_out/${tag}_traffic_${wl}.txt; mkdir -p gpurun_out; : > $out
dbs=""
for set in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmc_step_$RANDOM; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/step_forward.py" $wl 3 > /tmp/log_step 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -2 /tmp/log_step >> $out; continue; }
  dbs="$dbs $f"
done
python tools/step_pmc_summary.py $wl 0.5 $dbs 2>&1 | cut -c1-130 >> $out
cat $out

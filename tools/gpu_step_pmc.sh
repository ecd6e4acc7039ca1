#!/bin/bash
# PMC counters of EVERY kernel of one step (tools/step_forward.py: the eager forward on one stream), summed over the step and
# set against what the chip offers during one step in flight:  bash tools/gpu_step_pmc.sh <tag> <workload> <ms_per_step_in_flight>
# -> gpurun_out/<tag>_pmc_step_<workload>.txt.  Per-dispatch counters serialise the kernels, so the utilisation of the
# pipelined steady state is DERIVED: sum over the step's kernels / (in-flight step time x chip capacity).
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
tag=${1:-r04}; wl=${2:-local}; ms=${3:-0.312}
out=gpurun_out/${tag}_pmc_step_${wl}.txt; mkdir -p gpurun_out; : > $out
dbs=""
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmc_step_$RANDOM; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/step_forward.py" $wl 3 > /tmp/log_step 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -z "$f" ] && { echo "no db for: $set" >> $out; tail -2 /tmp/log_step >> $out; continue; }
  dbs="$dbs $f"
done
python tools/step_pmc_summary.py $wl $ms $dbs >> $out
cat $out

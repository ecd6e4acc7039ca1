#!/bin/bash
# SQ counters of the batched FPS (tools/fps_pmc.py), three separate --pmc passes: bash tools/gpu_pmc_fps.sh <tag>
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
tag=${1:-r02}
out=gpurun_out/${tag}_pmc_fps.txt
echo "rocprofv3 --pmc <SQ counters> --kernel-trace on tools/fps_pmc.py (ordered FPS 8 x (8192 -> 1024), fps_batched_kernel<8,16>), separate passes; tools/rocpd_summary.py --pmc" > $out
for set in "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"; do
  d=/tmp/pmc_fps_$RANDOM; rm -rf $d
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/fps_pmc.py" > /tmp/log_fpspmc 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_summary.py --pmc $f | grep -E "kernel |fps_batched" >> $out
done
cat $out

#!/bin/bash
# SQ / memory counters of the persistent flex_conv (tools/flex_conv_pmc.py), separate --pmc passes: bash tools/gpu_pmc_x6.sh <tag>
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
tag=${1:-r02}
out=gpurun_out/${tag}_pmc_flex_conv_x6.txt
echo "rocprofv3 --pmc <counters> --kernel-trace on tools/flex_conv_pmc.py (flex_conv_x6 64->64, B=8, N=8192, K=8; 24 launches), separate passes; tools/rocpd_summary.py --pmc" > $out
for set in "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmc_x6_$RANDOM; rm -rf $d
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace -d $d -o p -- env PYTHONPATH="$root" python "$root/tools/flex_conv_pmc.py" > /tmp/log_x6pmc 2>&1)
  f=$(find $d -name "*.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_summary.py --pmc $f | grep -E "kernel |flex_conv_x6" >> $out
done
cat $out

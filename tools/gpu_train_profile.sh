#!/bin/bash
# kernel-trace summary of the training step (bash tools/gpu_train_profile.sh <tag>) -> gpurun_out/<tag>_kernel_stats_train.txt
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
export TMPDIR=/tmp
tag=${1:-r03_x}
mkdir -p gpurun_out
d=/tmp/prof_train; rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o p -- python "$root/bench.py" --workload train --steps 10 --warmup 3 --no-cpu-baseline > /tmp/log_train 2>&1)
f=$(find $d -name "*.db" | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python bench.py --workload train --steps 10 --warmup 3   (MI355X; tools/rocpd_summary.py)"; echo "bench line of the same (profiled) run:"; grep '^{"metric"' /tmp/log_train; echo; python tools/rocpd_summary.py $f | head -80; } > gpurun_out/${tag}_kernel_stats_train.txt
python tools/fills_in_step.py $f > gpurun_out/${tag}_fills_in_step.txt 2>&1

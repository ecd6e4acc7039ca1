#!/bin/bash
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/prof_train; rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o p -- python "$root/bench.py" --workload train --steps 10 --warmup 3 > /tmp/log_train 2>&1)
f=$(find $d -name "*.db" | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python bench.py --workload train --steps 10 --warmup 3"; grep '^{"metric"' /tmp/log_train; python tools/rocpd_summary.py $f | head -70; } > gpurun_out/r02_train_kernel_stats.txt
head -64 gpurun_out/r02_train_kernel_stats.txt | cut -c1-160

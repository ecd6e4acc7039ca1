#!/bin/bash
# timeline of one graph replay of a workload: bash tools/gpu_timeline.sh local|global|cfg5
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
w=${1:-local}
d=/tmp/prof_tl_$w; rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o p -- python "$root/bench.py" --workload $w --no-cpu-baseline --no-extras > /tmp/log_tl 2>&1)
f=$(find $d -name "*.db" | head -1)
key="spatial_sort_kernel<8>"; [ $w = global ] && key="spatial_sort_kernel<4>"; [ $w = cfg5 ] && key="spatial_sort_kernel<16>"
grep '^{"metric"' /tmp/log_tl | cut -c1-200
python tools/timeline.py $f "$key" | sed -n 3,40p | cut -c1-150

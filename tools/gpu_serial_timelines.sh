#!/bin/bash
# timelines of ONE replay with one step at a time (bench.py --inflight 1): bash tools/gpu_serial_timelines.sh <tag>
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$root/gpurun_out"
for w in local global; do
  d=/tmp/prof_s_$w; rm -rf "$d"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$d" -o p -- python "$root/bench.py" --workload $w --inflight 1 --no-cpu-baseline --no-extras --no-streaming --repeats 0 > /tmp/log_s_$w 2>&1)
  f=$(find "$d" -name "*.db" | head -1)
  key="spatial_sort_kernel<8>"; [ $w = global ] && key="spatial_sort_kernel<4>"
  { echo "one step at a time (bench.py --workload $w --inflight 1 --no-extras), timeline of one graph replay (tools/timeline.py):"; grep -o '"ms_per_step": [0-9.]*' /tmp/log_s_$w | head -1; python "$root/tools/timeline.py" "$f" "$key" | sed -n 1,60p; } > "$root/gpurun_out/${tag}_timeline_serial_$w.txt"
done

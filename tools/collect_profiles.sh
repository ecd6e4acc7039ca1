#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/collect_profiles.sh <tag>
# rocprofv3 --kernel-trace --stats of `python bench.py` for both workloads -> gpurun_out/<tag>_kernel_stats_<workload>.txt
# (per-kernel table, the bench line of the same run, the timeline of one graph replay).  Copy into profiles/ afterwards.
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$root/gpurun_out"
for w in local global; do
  d=/tmp/prof_${tag}_$w
  rm -rf "$d"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$d" -o p -- python "$root/bench.py" --workload $w --no-cpu-baseline > /tmp/log_$w 2>&1)
  f=$(find "$d" -name "*.db" | head -1)
  out="$root/gpurun_out/${tag}_kernel_stats_$w.txt"
  key="spatial_sort_kernel<8>"; [ $w = global ] && key="spatial_sort_kernel<4>"
  {
    echo "rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --no-cpu-baseline   (MI355X; tools/rocpd_summary.py)"
    echo "bench line of the same (profiled) run:"
    grep '^{"metric"' /tmp/log_$w
    echo
    python "$root/tools/rocpd_summary.py" "$f"
    echo
    echo "timeline of one graph replay (tools/timeline.py): start offset, duration, hardware queue"
    python "$root/tools/timeline.py" "$f" "$key" | sed -n 3,45p
  } > "$out"
  echo "wrote $out"
done

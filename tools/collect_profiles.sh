#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/collect_profiles.sh <tag>
# rocprofv3 --kernel-trace --stats of `python bench.py --no-extras` for both workloads
#   -> gpurun_out/<tag>_kernel_stats_<workload>.txt  (per-kernel table, the bench line of the same run, the timeline of one
#      graph replay; --no-extras keeps the roofline / second-workload / two-in-flight measurements out of the table)
# and of the roofline kernel alone (tools/flex_conv_pmc.py: flex_conv 64->64, B=8, N=8192, K=8)
#   -> gpurun_out/<tag>_kernel_stats_flex_conv.txt  (its average duration is what bench.py's roofline must agree with).
# Copy into profiles/ afterwards.
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$root/gpurun_out"
for w in local global; do
  d=/tmp/prof_${tag}_$w
  rm -rf "$d"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$d" -o p -- python "$root/bench.py" --workload $w --no-cpu-baseline --no-extras --no-streaming --repeats 0 > /tmp/log_$w 2>&1)
  f=$(find "$d" -name "*.db" | head -1)
  out="$root/gpurun_out/${tag}_kernel_stats_$w.txt"
  key="spatial_sort_kernel<8>"; [ $w = global ] && key="spatial_sort_kernel<4>"
  {
    echo "rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --no-cpu-baseline --no-extras   (MI355X; tools/rocpd_summary.py)"
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
d=/tmp/prof_${tag}_flex
rm -rf "$d"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$d" -o p -- env PYTHONPATH="$root" python "$root/tools/flex_conv_pmc.py" > /tmp/log_flex 2>&1)
f=$(find "$d" -name "*.db" | head -1)
out="$root/gpurun_out/${tag}_kernel_stats_flex_conv.txt"
{
  echo "rocprofv3 --kernel-trace --stats -- python tools/flex_conv_pmc.py   (the roofline kernel alone: flex_conv 64->64, B=8, N=8192, K=8)"
  python "$root/tools/rocpd_summary.py" "$f"
} > "$out"
echo "wrote $out"

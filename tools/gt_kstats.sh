#!/bin/bash
# Run ON THE GPU BOX from the repo root: per-kernel durations of the global tail alone (tools/gt_bench.py under
# rocprofv3 --kernel-trace --stats).  Usage: bash tools/gt_kstats.sh [lib.so ...]   (default: the product library)
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=("")
for lib in "${libs[@]}"; do
  d=/tmp/prof_gt_$$; rm -rf "$d"
  (cd /tmp && DH3D_HIP_LIB=${lib:+$root/$lib} rocprofv3 --kernel-trace --stats -d "$d" -o p -- env PYTHONPATH="$root" python "$root/tools/gt_bench.py" > /tmp/log_gt 2>&1)
  echo "== ${lib:-product}: $(grep -a global_tail /tmp/log_gt | tail -1)"
  python "$root/tools/rocpd_summary.py" "$(find "$d" -name '*.db' | head -1)" | grep -E 'interp_head|netvlad|linear_x6|Name|name' | cut -c1-150
done

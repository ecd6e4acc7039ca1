#!/bin/bash
# GPU box: bash tools/gpu_inflight_overlap.sh [local|global] -- steady-state overlap statistics of the in-flight pipeline
root=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
w=${1:-local}; d=/tmp/prof_ov_$w; rm -rf "$d"
(cd /tmp && rocprofv3 --kernel-trace -d "$d" -o p -- python "$root/bench.py" --workload $w --steps 200 --no-cpu-baseline --no-extras --no-streaming --repeats 0 > /tmp/log_ov 2>&1)
grep -o '"ms_per_step": [0-9.]*' /tmp/log_ov | head -1
gap=400; [ $w = global ] && gap=600
python "$root/tools/inflight_overlap.py" "$(find "$d" -name '*.db' | head -1)" $gap

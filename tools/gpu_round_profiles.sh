#!/bin/bash
# per-round evidence run (bash tools/gpu_round_profiles.sh r03_b): whole GPU suite, bench, kernel-trace summaries (local / global / flex_conv / train)
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
export TMPDIR=/tmp
tag=${1:-r03_a}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
bash tools/collect_profiles.sh $tag > /dev/null 2>&1
d=/tmp/prof_train; rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o p -- python "$root/bench.py" --workload train --steps 10 --warmup 3 --no-cpu-baseline > /tmp/log_train 2>&1)
f=$(find $d -name "*.db" | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python bench.py --workload train --steps 10 --warmup 3   (MI355X; tools/rocpd_summary.py)"; echo "bench line of the same (profiled) run:"; grep '^{"metric"' /tmp/log_train; echo; python tools/rocpd_summary.py $f | head -80; } > gpurun_out/${tag}_kernel_stats_train.txt
python tools/fills_in_step.py $f > gpurun_out/${tag}_fills_in_step.txt 2>&1
ls -la gpurun_out | tail -12

#!/bin/bash
# round-5 closing evidence (reduced: what changed since r05_z is the global walk): bash tools/gpu_round5_final.sh <tag>
#   smoke, the whole GPU suite, the default bench line (+ side file), the global workload's line, kernel-trace summaries
#   (local / global / flex_conv), serial timelines
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
export TMPDIR=/tmp
tag=${1:-r05_zz}
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${tag}_smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; wc -c gpurun_out/${tag}_bench.json
cp gpurun_out/bench_extras.json gpurun_out/${tag}_bench_sidefile.json
timeout 600 python bench.py --workload global --no-cpu-baseline --extras-file gpurun_out/${tag}_bench_global_sidefile.json > gpurun_out/${tag}_bench_global.json 2>> gpurun_out/${tag}_bench.err
bash tools/collect_profiles.sh $tag > /dev/null 2>&1
bash tools/gpu_serial_timelines.sh $tag > /dev/null 2>&1
ls -la gpurun_out | grep $tag
cat gpurun_out/${tag}_bench.json | cut -c1-700
cat gpurun_out/${tag}_bench_global.json | cut -c1-400

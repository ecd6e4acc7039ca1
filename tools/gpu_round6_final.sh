#!/bin/bash
# round-6 closing evidence: bash tools/gpu_round6_final.sh <tag>   (GPU box, from the repo root)
#   smoke, the whole GPU suite, the default bench line (+ side file), the --extras record, kernel-trace summaries
#   (local / global / flex_conv), serial timelines, per-kernel HBM traffic of both steps, in-flight overlap statistics
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
export TMPDIR=/tmp
tag=${1:-r06_z}
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${tag}_smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; wc -c gpurun_out/${tag}_bench.json
cp gpurun_out/bench_extras.json gpurun_out/${tag}_bench_sidefile.json
timeout 1500 python bench.py --extras --extras-file gpurun_out/${tag}_bench_extras_sidefile.json > gpurun_out/${tag}_bench_extras.json 2>> gpurun_out/${tag}_bench.err; echo "extras rc=$?"
bash tools/collect_profiles.sh $tag > /dev/null 2>&1
bash tools/gpu_serial_timelines.sh $tag > /dev/null 2>&1
bash tools/gpu_step_traffic.sh $tag local > /dev/null 2>&1
bash tools/gpu_step_traffic.sh $tag global > /dev/null 2>&1
{ bash tools/gpu_inflight_overlap.sh local; bash tools/gpu_inflight_overlap.sh global; } > gpurun_out/${tag}_inflight_overlap.txt 2>&1
ls -la gpurun_out | grep $tag
cat gpurun_out/${tag}_bench.json | cut -c1-900

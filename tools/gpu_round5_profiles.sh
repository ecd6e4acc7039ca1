#!/bin/bash
# round-5 evidence run: bash tools/gpu_round5_profiles.sh <tag>
#   whole GPU suite, the default bench line (+ its side file), the --extras record, kernel-trace summaries (local /
#   global / flex_conv / train / train_local), serial timelines, per-step PMC (local + global), the co-issue probe
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
export TMPDIR=/tmp
tag=${1:-r05_z}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; wc -c gpurun_out/${tag}_bench.json
cp gpurun_out/bench_extras.json gpurun_out/${tag}_bench_sidefile.json
timeout 1200 python bench.py --extras --extras-file gpurun_out/${tag}_bench_extras.json > gpurun_out/${tag}_bench_with_extras.json 2>> gpurun_out/${tag}_bench.err; echo "bench --extras rc=$?"
timeout 600 python bench.py --workload global --no-cpu-baseline --extras-file gpurun_out/${tag}_bench_global_sidefile.json > gpurun_out/${tag}_bench_global.json 2>> gpurun_out/${tag}_bench.err
bash tools/collect_profiles.sh $tag > /dev/null 2>&1
for w in train train_local; do
  d=/tmp/prof_$w; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o p -- python "$root/bench.py" --workload $w --steps 10 --warmup 3 --no-cpu-baseline --extras-file /dev/null > /tmp/log_$w 2>&1)
  f=$(find $d -name "*.db" | head -1)
  { echo "rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --steps 10 --warmup 3   (MI355X; tools/rocpd_summary.py)"; echo "bench line of the same (profiled) run:"; grep '^{"metric"' /tmp/log_$w | cut -c1-1500; echo; python tools/rocpd_summary.py $f | head -70; } > gpurun_out/${tag}_kernel_stats_$w.txt
done
bash tools/gpu_serial_timelines.sh $tag > /dev/null 2>&1
ms_l=$(python -c "import json;d=[json.loads(l) for l in open('gpurun_out/${tag}_bench.json') if l.startswith('{')][-1];print('%.4f'%d['ms_per_step'])")
ms_g=$(python -c "import json;d=[json.loads(l) for l in open('gpurun_out/${tag}_bench_global.json') if l.startswith('{')][-1];print('%.4f'%d['ms_per_step'])")
bash tools/gpu_step_pmc.sh $tag local $ms_l > /dev/null 2>&1
bash tools/gpu_step_pmc.sh $tag global $ms_g > /dev/null 2>&1
bash tools/kernel_pmc.sh ${tag}_flex_conv_x6 "%flex_conv_x6_kernel%" tools/flex_conv_pmc.py > /dev/null 2>&1
./tools/coissue_probe2 > gpurun_out/${tag}_coissue_probe2.txt 2>&1
ls -la gpurun_out | grep $tag

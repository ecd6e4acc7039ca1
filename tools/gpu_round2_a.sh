#!/bin/bash
# first GPU call of round 2: new tests, then whole suite, bench, kernel-trace profile
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/a_smoke.log 2>&1; echo "smoke rc=$?" 
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=900 -k "parity_fullsize or losses or layers or training or fps_mode or fps_modes or config_reachable" > gpurun_out/a_newtests.log 2>&1; echo "newtests rc=$?"
tail -15 gpurun_out/a_newtests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/a_bench.json

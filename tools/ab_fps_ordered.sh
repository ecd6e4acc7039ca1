for cfg in "1 1" "1 0" "0 0" "1 1"; do set -- $cfg; for wl in local global; do
echo -n "ORDERED=$1 GRID=$2 $wl: "; DH3D_FPS_ORDERED=$1 DH3D_SAMPLED_GRID=$2 python bench.py --workload $wl --no-extras --no-cpu-baseline --no-streaming --no-global-line --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight %.4f ms (%.0f)  serial %.4f ms' % (d['ms_per_step'], d['value'], d['one_step_at_a_time']['ms_per_step']))"
done; done

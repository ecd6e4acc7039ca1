# usage: bash tools/ab_env2.sh "VAR=a VAR2=b" "VAR=c" ...   (each argument one environment; local + global, serial + in flight)
for rep in 1 2; do for cfg in "$@"; do for wl in local global; do
echo -n "[$cfg] $wl: "; env $cfg python bench.py --workload $wl --no-extras --no-cpu-baseline --no-streaming --no-global-line --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight %.4f ms (%.0f)  serial %.4f ms' % (d['ms_per_step'], d['value'], d['one_step_at_a_time']['ms_per_step']))"
done; done; done

"""Dev: A/B of an environment switch on the serial and in-flight steps, one fresh process per setting.
usage: python tools/ab_env.py DH3D_KNN_GRID 0 1"""
import json, os, subprocess, sys
var, vals = sys.argv[1], sys.argv[2:]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for rep in range(2):
    for v in vals:
        for wl in ("local", "global"):
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", wl, "--no-extras", "--no-cpu-baseline",
                                  "--repeats", "2"], env=dict(os.environ, **{var: v}), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            d = json.loads([l for l in out.splitlines() if l.startswith('{"metric"')][-1])
            print("%s=%s %-6s in flight %.0f (%.4f ms)  serial %.0f (%.4f ms)" % (var, v, wl, d["value"], d["ms_per_step"],
                  d["one_step_at_a_time"]["value"], d["one_step_at_a_time"]["ms_per_step"]), flush=True)

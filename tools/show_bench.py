"""Dev: print the headline numbers and the roofline_step tables of a bench.py JSON line.  usage: python tools/show_bench.py file"""
import json, sys
d = None
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l)
def show(rs):
    if not rs or 'error' in rs:
        print(rs); return
    for r in rs['kernels']:
        print("  %-40s x%d floor %7.2f (hbm %6.2f f32 %6.2f) meas %7.2f frac %.3f" % (r['kernel'][:40], r['launches'], r['floor_us'], r['floor_hbm_us'], r['floor_f32_us'], r['measured_us'], r['frac'] or 0.0))
    print("  ", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rs.items() if k not in ('kernels', 'note', 'peaks', 'workload', 'unlisted_kernel_names')})
print("value %.0f  ms %.4f | serial %.0f ms %.4f" % (d['value'], d['ms_per_step'], d['one_step_at_a_time']['value'], d['one_step_at_a_time']['ms_per_step']))
if 'repeats' in d: print("  repeats in flight:", d['repeats']['ms_per_step'])
print("roofline frac %.4f launch_ms %.4f" % (d.get('roofline', {}).get('frac', 0), d.get('roofline', {}).get('launch_ms', 0)))
show(d.get('roofline_step'))
for o in d.get('other_workloads', []):
    print(o['key'], "value %.0f ms %.4f" % (o['value'], o['ms_per_step']), o.get('one_step_at_a_time'))
    if 'roofline_step' in o: show(o['roofline_step'])
    if 'kernel_roofline' in o: print("   cfg5 kernel:", o['kernel_roofline']['launch_ms'], o['kernel_roofline']['strict_hbm']['frac'])
print(d.get('throughput_by_steps_in_flight'))

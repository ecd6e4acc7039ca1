"""Sums the per-dispatch PMC counters of the LAST forward in each rocprofv3 database (tools/gpu_step_pmc.sh) per kernel
and over the step, and relates the step totals to the chip's capacity during one step in flight."""
import re, sqlite3, sys
wl, ms = sys.argv[1], float(sys.argv[2])
per = {}   # kernel -> counter -> sum over the last forward
order = []
for path in sys.argv[3:]:
    c = sqlite3.connect(path)
    rows = c.execute("select name, counter_name, counter_value, start from pmc_events order by start").fetchall()
    starts = [r[3] for r in rows if "spatial_sort_kernel" in r[0] and "<1>" not in r[0]]
    t0 = max(starts) if starts else 0
    for name, cn, val, st in rows:
        if st < t0:
            continue
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", name)
        n = re.sub(r"\(.*", "", n)[:58]
        if n not in per:
            per[n] = {}
            order.append(n)
        per[n][cn] = per[n].get(cn, 0.0) + val
counters = sorted({cn for d in per.values() for cn in d})
print("PMC per kernel, summed over the launches of ONE %s step (eager, one stream); SQ_* cycle counters in quad-cycles "
      "except SQ_BUSY_CU_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES (cycles); FETCH/WRITE_SIZE raw KB" % wl)
print("%-58s " % "kernel" + " ".join("%16s" % c[-16:] for c in counters))
tot = {}
for n in order:
    print("%-58s " % n + " ".join("%16.0f" % per[n].get(c, 0.0) for c in counters))
    for c in counters:
        tot[c] = tot.get(c, 0.0) + per[n].get(c, 0.0)
print("%-58s " % "STEP TOTAL" + " ".join("%16.0f" % tot.get(c, 0.0) for c in counters))
# capacity of the chip during one step in flight
clk = 2.0e9   # ~ effective shader clock under load (GRBM_GUI_ACTIVE / wall of the profiled kernels is 1.9-2.1 GHz)
cu_cycles = ms * 1e-3 * clk * 256
print()
print("one step in flight = %.3f ms -> %.3e CU-cycles at %.1f GHz x 256 CUs" % (ms, cu_cycles, clk / 1e9))
def pct(x): return "%.1f %%" % (100.0 * x)
if "SQ_BUSY_CU_CYCLES" in tot: print("  CU busy (some wave resident):      ", pct(tot["SQ_BUSY_CU_CYCLES"] / cu_cycles), " (SQ_BUSY_CU_CYCLES is summed over shader engines: relative reading)")
if "SQ_ACTIVE_INST_VALU" in tot: print("  VALU issue, per SIMD:              ", pct(tot["SQ_ACTIVE_INST_VALU"] * 4 / (cu_cycles * 4)))
if "SQ_VALU_MFMA_BUSY_CYCLES" in tot: print("  matrix pipe busy, per SIMD:        ", pct(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (cu_cycles * 4)))
if "SQ_ACTIVE_INST_LDS" in tot: print("  LDS instruction issue, per CU:     ", pct(tot["SQ_ACTIVE_INST_LDS"] * 4 / cu_cycles))
if "SQ_ACTIVE_INST_VMEM" in tot: print("  vector-memory issue, per CU:       ", pct(tot["SQ_ACTIVE_INST_VMEM"] * 4 / cu_cycles))
if "SQ_WAVE_CYCLES" in tot and "SQ_WAIT_ANY" in tot:
    print("  of all wave cycles: waiting %s, issuing %s" % (pct(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"]), pct(tot.get("SQ_ACTIVE_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"])))
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    b = (tot["FETCH_SIZE"] * 2 + tot["WRITE_SIZE"]) * 1024
    print("  HBM traffic %.1f MB per step (FETCH x2 gfx950 correction + WRITE) -> %.0f GB/s = %s of 8 TB/s" % (b / 1e6, b / (ms * 1e-3) / 1e9, pct(b / (ms * 1e-3) / 8e12)))

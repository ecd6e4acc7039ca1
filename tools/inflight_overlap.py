"""Dev: how the steady state of an in-flight bench run uses the chip (rocprofv3 rocpd db): over the timed region, the share of
time with 0 / 1 / 2 / 3+ chip-wide kernels running (everything except the one-CU-per-cloud kernels: FPS, the sort)."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
fps = [r for r in rows if "fps_list_kernel" in r[0]]
# the steady state of the in-flight region: the longest run of FPS launches that start less than GAP (0.4 ms; global: 0.6) apart (one step at a
# time they are a step apart, and the clock-ramp replays in front of every timed region are synchronised one by one)
starts = [r[1] for r in fps]
GAP = int(float(sys.argv[2]) * 1000) if len(sys.argv) > 2 else 400000   # ns; argv[2] in us
best, cur = (0, 0), 0
for i in range(1, len(starts)):
    if starts[i] - starts[i - 1] > GAP:
        cur = i
    if i - cur > best[1] - best[0]:
        best = (cur, i)
n = best[1] - best[0]
t0, t1 = starts[best[0] + n // 4], starts[best[1] - n // 4]
narrow = ("fps_list_kernel", "spatial_sort_kernel")
win = [r for r in rows if t0 <= r[1] < t1 and not any(k in r[0] for k in narrow)]
ev = sorted([(r[1], 1) for r in win] + [(min(r[2], t1), -1) for r in win])
hist, depth, last = {}, 0, t0
for t, d in ev:
    hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
    depth += d; last = t
tot = float(t1 - t0)
nsteps = sum(1 for r in fps if t0 <= r[1] < t1)
print("window %.1f ms, %d steps (%.1f us per step); chip-wide kernels running: none %.1f %%, one %.1f %%, two %.1f %%, three or more %.1f %%" % (
    tot / 1e6, nsteps, tot / 1e3 / max(nsteps, 1), *(100.0 * hist.get(k, 0) / tot for k in range(4))))
print("sum of chip-wide kernel durations per step: %.1f us" % (sum(r[2] - r[1] for r in win) / 1e3 / max(nsteps, 1)))

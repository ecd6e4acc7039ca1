"""Dev: kernel timeline of the steady state of an in-flight bench run (rocprofv3 rocpd db): a 1.2 ms window in the middle of
the timed region, every kernel with start offset, duration, queue; and how busy the window is (union of kernel intervals,
sum of durations)."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else "stream_id" if "stream_id" in cols else "0"
rows = c.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
fps = [r for r in rows if "fps_list_kernel" in r[0]]
mid = fps[len(fps) * 3 // 4][1]
win = [r for r in rows if mid <= r[1] < mid + 1200000]
ev = sorted([(r[1], 1) for r in win] + [(r[2], -1) for r in win])
busy, depth, last = 0, 0, None
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
print("window 1200 us: %d kernels, union busy %.0f us, sum of durations %.0f us, fps launches %d" %
      (len(win), busy / 1e3, sum(r[2] - r[1] for r in win) / 1e3, sum(1 for r in win if "fps_list" in r[0])))
for n, s, e, qq in win[:int(sys.argv[2]) if len(sys.argv) > 2 else 80]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)[:44]
    print("%8.1f +%7.1f q%-2s %s" % ((s - mid) / 1e3, (e - s) / 1e3, qq, n))

"""Print every kernel of a time window in the middle of a rocprofv3 rocpd db (start offset, duration, queue): what
overlaps what when several graph instances are in flight.   python tools/timeline_window.py p.db [window_us] [key]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 800.0
key = sys.argv[3] if len(sys.argv) > 3 else "spatial_sort_kernel<8>"
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else "stream_id" if "stream_id" in cols else "0"
rows = c.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
idx = [i for i, r in enumerate(rows) if key in r[0]]
i0 = idx[len(idx) * 3 // 4]          # a step start well inside the timed loop
t0 = rows[i0][1]
busy = {}
for n, s, e, qq in rows[i0:]:
    if (s - t0) / 1e3 > win: break
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)[:52]
    print("%8.1f us  +%7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qq, n))

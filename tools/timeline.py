"""Print the kernel timeline of the last graph replay in a rocprofv3 rocpd db (start offset, duration, stream/queue)."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end, %s from kernels order by start" % ("queue_id" if "queue_id" in cols else "stream_id" if "stream_id" in cols else "0")).fetchall()
# find last occurrence of spatial_sort_kernel<8> (local step start) -> print until l2norm
names = [r[0] for r in rows]
key = sys.argv[2] if len(sys.argv) > 2 else "spatial_sort_kernel<8>"
idx = [i for i, n in enumerate(names) if key in n]
i0 = idx[-2] if len(idx) > 1 else idx[-1]
t0 = rows[i0][1]
for n, s, e, q in rows[i0 - 2:i0 + 40]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)[:60]
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))

"""Which fills / copies does one replayed training step contain?  python tools/fills_in_step.py <rocpd .db>
Prints, for the last complete step in the trace, every fill / copy kernel with its duration and its neighbours."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
rows = c.execute("select %s, start, end from kernels order by start" % namecol).fetchall()
names = [re.sub(r"\(anonymous namespace\)::|^void ", "", r[0])[:60] for r in rows]
# a step = from one quadruplet_loss kernel to the next
marks = [i for i, n in enumerate(names) if n.startswith("quadruplet_loss")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 3   # (bench.py --workload train: plain steps first, then the sharded-path ones)
a, b = marks[k], marks[k + 1]
print("step of %d kernels, %.1f us" % (b - a, (rows[b][1] - rows[a][1]) / 1e3))
for i in range(a, b):
    if "fillBuffer" in names[i] or "copyBuffer" in names[i] or "FillFunctor" in names[i]:
        print("%8.2f us  %-45s after %-45s before %s" % ((rows[i][2] - rows[i][1]) / 1e3, names[i], names[i - 1][:45], names[i + 1][:45]))
print()
print("every kernel of the step, in start order:")
for i in range(a, b):
    print("%9.2f %8.2f us  %s" % ((rows[i][1] - rows[a][1]) / 1e3, (rows[i][2] - rows[i][1]) / 1e3, names[i]))

#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats output) as a per-kernel table.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 90 else name[:87] + "..."


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by %s order by 3 desc" % (namecol, namecol)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-90s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for n, cnt, tot, avg, mn, mx in rows:
        print("%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (short(n), cnt, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                             100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])

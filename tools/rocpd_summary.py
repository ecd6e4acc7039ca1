#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats output) as a per-kernel table.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt
    python tools/rocpd_summary.py --pmc gpurun_out/pmc/x_results.db      (per-kernel counter values of a --pmc pass)
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


def pmc(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, counter_name, count(*), avg(counter_value), min(counter_value), max(counter_value), "
                     "avg(duration) from pmc_events group by name, counter_name order by 2, 4 desc").fetchall()
    print("%-70s %-28s %6s %14s %14s %14s %10s" % ("kernel", "counter", "calls", "avg", "min", "max", "avg_dur_us"))
    for n, cn, cnt, avg, mn, mx, dur in rows:
        print("%-70s %-28s %6d %14.1f %14.1f %14.1f %10.2f" % (short(n)[:70], cn, cnt, avg, mn, mx, dur / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "--pmc":
        pmc(sys.argv[2])
    else:
        main(sys.argv[1])

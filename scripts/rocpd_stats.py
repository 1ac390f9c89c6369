#!/usr/bin/env python
"""Per-kernel statistics (the `rocprofv3 --kernel-trace --stats` summary) out of the rocpd SQLite
database rocprofv3 7.2 writes by default.  usage: rocpd_stats.py results.db [> profiles/...txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-72s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for name, calls, tot, avg, mn, mx in rows:
        short = name.split("(")[0][-72:]
        print("%-72s %8d %14d %12.1f %12d %12d %6.2f%%" % (short, calls, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])

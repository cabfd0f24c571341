"""per-kernel summary of a rocprofv3 --kernel-trace database: python tools/kstats.py <results.db> [name filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name order by 3 desc").fetchall()
for name, calls, tot, avg, mn in rows:
    if flt in name:
        print("%-90s calls %5d total %10.3f ms avg %9.1f us min %9.1f us" % (name[:90], calls, tot / 1e6, avg / 1e3, mn / 1e3))

#!/usr/bin/env python
"""Per-kernel summary (calls, total ms, average ms, share) of a rocprofv3 --kernel-trace results.db as CSV.
usage: python tools/kernel_stats.py <results.db> [header comment]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc"))
tot = sum(r[2] for r in rows)
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print("name,calls,total_ms,avg_ms,pct")
for name, n, s, a in rows:
    print('"%s",%d,%.3f,%.4f,%.3f' % (name, n, s / 1e6, a / 1e6, 100.0 * s / tot))

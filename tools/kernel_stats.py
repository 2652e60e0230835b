#!/usr/bin/env python
"""Per-kernel summary (calls, total ms, average ms, share) of a rocprofv3 --kernel-trace results.db as CSV.
Conv / chain / weight-gradient launches dispatched BEFORE the first plan execution (= before the first `hvn_conv0` kernel) are the
engines' launch-shape autotune (`Engine.autotune_tiles`, `TrainEngine.autotune_tiles`: every re-tileable shape timed with its
candidates at engine build); they are reported on their own comment line and left out of the table, so that
`total_ms / plan executions` is the per-step time again.
usage: python tools/kernel_stats.py <results.db> [header comment]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
first = c.execute("select min(start) from kernels where name like '%hvn_conv0%'").fetchone()[0]
tune = (0, 0)
cond = ""
if first is not None:
    tune = c.execute("select count(*), coalesce(sum(duration), 0) from kernels where start < ? and (name like '%hvn_conv_igemm%' or name like '%hvn_conv_chain%' or name like '%hvn_conv_wgrad%')", (first,)).fetchone()
    cond = "where not (start < %d and (name like '%%hvn_conv_igemm%%' or name like '%%hvn_conv_chain%%' or name like '%%hvn_conv_wgrad%%'))" % first
rows = list(c.execute("select name, count(*), sum(duration), avg(duration) from kernels %s group by name order by sum(duration) desc" % cond))
tot = sum(r[2] for r in rows)
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
if tune[0]:
    print("# excluded: %d launch-shape autotune launches (conv / chain / wgrad) before the first plan execution, %.3f ms in total" % (tune[0], tune[1] / 1e6))
print("name,calls,total_ms,avg_ms,pct")
for name, n, s, a in rows:
    print('"%s",%d,%.3f,%.4f,%.3f' % (name, n, s / 1e6, a / 1e6, 100.0 * s / tot))

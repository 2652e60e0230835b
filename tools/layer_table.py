#!/usr/bin/env python
"""Per-layer conv table from a rocprofv3 --kernel-trace results.db of `python bench.py`:
maps the LAST plan execution's hvn_conv_igemm_f32 dispatches (in order) onto the plan's CONV ops.
usage: python tools/layer_table.py <results.db> [batch]"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd.plan import build_plan  # noqa: E402
from hover_net_amd.synth import synth_state_dict  # noqa: E402

db, batch = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = build_plan(synth_state_dict("original", 5, seed=0), "original", 5)
convs = [o for o in P.ops if o.kind == 2]
c = sqlite3.connect(db)
rows = list(c.execute("select name,duration,grid_x from kernels where name like '%igemm%' order by start"))
last = rows[-len(convs):]
tot, agg = 0, {}
print("%-44s %-22s %9s %8s" % ("op", "shape", "us", "TFLOP/s"))
for o, (name, dur, gx) in zip(convs, last):
    fl = o.flops() * batch
    tot += dur
    key = o.name.split(".units")[0] if "units" in o.name else o.name
    a = agg.setdefault(key, [0, 0])
    a[0] += dur
    a[1] += fl
    print("%-44s k%dx%d s%d %4d->%-4d @%-3d %9.1f %8.1f" % (o.name, o.kh, o.kw, o.stride, o.x.c, o.cout, o.y.h, dur / 1e3, fl / dur / 1e3))
print("total conv ms %.2f" % (tot / 1e6))
for k, (d, f) in agg.items():
    print("%-30s %8.2f ms %6.1f TFLOP/s" % (k, d / 1e6, f / d / 1e3))

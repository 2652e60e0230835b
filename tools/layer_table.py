#!/usr/bin/env python
"""Per-op table from a rocprofv3 --kernel-trace results.db of `python bench.py`: maps the LAST plan execution's
network-kernel dispatches (bench.py's single-stream roofline pass, one launch per plan op, in order) onto the plan's
ops.  TFLOP/s columns: algorithmic (direct-convolution FLOPs) and executed (after Winograd); for launches on the bf16x3 kernel
(shape marked `x3`) "executed" counts each fp32 product once (fp32-equivalent: x 6 = the bf16 MFMA FLOPs the launch issues).
usage: python tools/layer_table.py <results.db> [batch]"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd.plan import build_plan  # noqa: E402
from hover_net_amd.synth import synth_state_dict  # noqa: E402

db, batch = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = build_plan(synth_state_dict("original", 5, seed=0), "original", 5)
c = sqlite3.connect(db)
pat = ("igemm", "hvn_conv_chain", "hvn_dense_grouped", "hvn_conv0", "hvn_upadd", "hvn_head", "hvn_predmap", "hvn_wino")
rows = [r for r in c.execute("select name,duration from kernels order by start") if any(p in r[0] for p in pat)]
last = rows[-len(P.ops):]
tot, agg = 0, {}
print("%-62s %-24s %9s %8s %8s" % ("op", "shape", "us", "algo TF", "exec TF"))
for o, (name, dur) in zip(P.ops, last):
    fl = o.flops() * batch if o.kind in (2, 8) else 0.0
    ex = o.extra.get("exec_flops", o.flops()) * batch if o.kind in (2, 8) else 0.0
    tot += dur
    key = o.name.split(".units")[0] if "units" in o.name else o.name
    key = key.split(".wino_")[0]
    a = agg.setdefault(key, [0, 0, 0])
    a[0] += dur
    a[1] += fl
    a[2] += ex
    shape = ("k%dx%d s%d %4d->%-4d @%-3d" % (o.kh, o.kw, o.stride, o.x.c, o.cout, o.y.h) + (" x3" if o.extra.get("x3") else "")) if o.kind == 2 else (
        "1x1 %d->%d->%d @%d" % (o.extra["cin_real"], o.cout, o.extra["cout2"], o.y.h) if o.kind == 8 else name.split("(")[0][:24])
    print("%-62s %-24s %9.1f %8.1f %8.1f" % (o.name, shape, dur / 1e3, fl / dur / 1e3, ex / dur / 1e3))
print("total network-kernel ms %.2f" % (tot / 1e6))
for k, (d, f, e) in agg.items():
    print("%-36s %8.2f ms %6.1f algo %6.1f exec TFLOP/s" % (k, d / 1e6, f / d / 1e3, e / d / 1e3))

#!/usr/bin/env python
"""HBM / fabric traffic per LAUNCH CLASS of one network step, next to each class's compulsory bytes.

Input: the two `rocprofv3 --pmc` passes bench.py makes for `roofline.traffic` (FETCH_SIZE, WRITE_SIZE: separate passes, one plan execution
on one launch stream each -- MI355X_MICROARCH.md, PMC slots) and the engine whose plan ran.  The hvn_* dispatches of the LAST plan execution
are matched to the plan's ops in launch order; FETCH_SIZE is doubled (the guide's gfx950 correction for wide coalesced reads), WRITE_SIZE
taken as reported.  Compulsory bytes of an op = every operand view read once + every output view written once + its weights (what a launch
that re-reads nothing would move; the network's own compulsory figure -- input tiles + logits -- is 0.23 GB).
`HVN_KEEP_PMC_TABLE=profiles/rNN_traffic_by_kernel.txt python bench.py` writes the table of that run."""
import sqlite3


def _dispatches(db, counter):
    c = sqlite3.connect(db)
    rows = list(c.execute("select dispatch_id, kernel_name, sum(value), min(start) from counters_collection where counter_name=? "
                          "group by dispatch_id order by min(start)", (counter,)))
    return [(r[1], float(r[2])) for r in rows if "hvn_" in r[1]]


def _view_bytes(v, batch, isz=4):
    return 0 if v is None else batch * v.h * v.w * v.c * isz


def op_class(op):
    """Launch class of a plan op: kernel family, and for the bf16x3 convolutions the reduction-length bucket."""
    k = op.kind
    if k == 1:
        return "conv0 (7x7 stem, uint8 in)"
    if k == 3:
        return "upadd"
    if k == 4:
        return "head"
    if k == 5:
        return "predmap"
    if k == 6:
        return "wino_in"
    if k == 7:
        return "wino_out"
    if k == 8:
        return "chain bf16x3 (conv3 + conv1)" if op.extra.get("x3") else "chain fp32 pipe"
    if int(op.extra.get("groups", 1)) > 1:
        return "dense grouped 5x5"
    if not op.extra.get("x3"):
        return "conv fp32 pipe"
    kred = op.kh * op.kw * (op.x.c + (op.extra["x2"].c if op.extra.get("x2") is not None else 0))
    name = "x3 Winograd-domain product" if op.extra.get("nbatch") else ("x3 1x1" if op.kh == 1 else "x3 %dx%d" % (op.kh, op.kw))
    return "%s, K %s" % (name, "<= 256" if kred <= 256 else ("<= 1024" if kred <= 1024 else "> 1024"))


def compulsory(op, batch):
    b = _view_bytes(op.x, batch) + _view_bytes(op.y, batch) + _view_bytes(op.res, batch) + _view_bytes(op.extra.get("x2"), batch)
    b += _view_bytes(op.extra.get("y2"), batch)
    if op.kind == 1:
        b = batch * op.x.h * op.x.w * 3 + _view_bytes(op.y, batch)      # uint8 image
    if op.kind == 4:
        b = _view_bytes(op.x, batch) + batch * op.y.h * op.y.w * op.y.c * 4
    for w in (op.w, op.extra.get("w2")):
        if w is not None and op.kind in (2, 8):
            b += w.size * (6 if op.extra.get("x3") else 4)             # bf16x3 launches read three bf16 planes per weight
    if op.kind == 2 and op.extra.get("nbatch"):
        nb = int(op.extra["nbatch"])
        b = batch * nb * op.x.w * (op.x.c + op.y.c) * 4 + op.w.size * (6 if op.extra.get("x3") else 4)
    return float(b)


def table(fetch_db, write_db, eng, batch):
    ops = list(eng.plan.ops)                 # one hvn_* dispatch per plan op, in launch order (single launch stream)
    f, w = _dispatches(fetch_db, "FETCH_SIZE"), _dispatches(write_db, "WRITE_SIZE")
    n = len(ops)
    f, w = f[-n:], w[-n:]
    if len(f) != n or len(w) != n:
        return "traffic table: expected %d hvn_* dispatches per plan execution, saw %d / %d\n" % (n, len(f), len(w))
    agg = {}
    for op, (kf, vf), (kw, vw) in zip(ops, f, w):
        c = op_class(op)
        a = agg.setdefault(c, [0, 0.0, 0.0, 0.0, set()])
        a[0] += 1
        a[1] += vf * 1024 * 2
        a[2] += vw * 1024
        a[3] += compulsory(op, batch)
        a[4].add(kf.split("(")[0].split("<")[0].replace("void ", "").strip())
    lines = ["# HBM / fabric bytes per launch class of ONE network step (batch %d), two rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE);" % batch,
             "# compulsory = each operand / output view once + weights.  GB = 1e9 bytes.",
             "%-36s %8s %10s %10s %10s %12s %7s  %s" % ("class", "launches", "fetch GB", "write GB", "total GB", "compulsory GB", "ratio", "kernels")]
    tot = [0, 0.0, 0.0, 0.0]
    for c, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        t = a[1] + a[2]
        lines.append("%-36s %8d %10.2f %10.2f %10.2f %12.2f %7.2f  %s" % (c, a[0], a[1] / 1e9, a[2] / 1e9, t / 1e9, a[3] / 1e9, t / max(a[3], 1.0), ", ".join(sorted(a[4]))))
        for i in range(4):
            tot[i] += a[i]
    t = tot[1] + tot[2]
    lines.append("%-36s %8d %10.2f %10.2f %10.2f %12.2f %7.2f" % ("ALL LAUNCHES", tot[0], tot[1] / 1e9, tot[2] / 1e9, t / 1e9, tot[3] / 1e9, t / max(tot[3], 1.0)))
    return "\n".join(lines) + "\n"

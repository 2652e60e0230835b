#!/usr/bin/env python
"""HBM traffic of the conv kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of
`python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline` (HVN_SPLIT=1 HVN_LANES=0).
Sums the counters over the conv dispatches (hvn_conv_igemm_f32 + hvn_dense_grouped*_f32) of the LAST plan execution.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by 2x.
usage: python tools/pmc_traffic.py <fetch.db> <write.db> <out.json>"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd.plan import build_plan  # noqa: E402
from hover_net_amd.synth import synth_state_dict  # noqa: E402


def total(db, counter, n_last):
    c = sqlite3.connect(db)
    rows = list(c.execute("select dispatch_id, sum(value), min(start) from counters_collection "
                          "where (kernel_name like '%igemm%' or kernel_name like '%conv_chain%' or kernel_name like '%dense_grouped%') and counter_name=? group by dispatch_id order by min(start)", (counter,)))
    rows = rows[-n_last:]
    return sum(r[1] for r in rows), len(rows)


def reduce(fetch_db, write_db, n):
    """HBM bytes of the last `n` conv dispatches from the two PMC passes (FETCH_SIZE / WRITE_SIZE are reported in KB)."""
    fetch_kb, nf = total(fetch_db, "FETCH_SIZE", n)
    write_kb, nw = total(write_db, "WRITE_SIZE", n)
    return {"launches": nf, "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb,
            "fetch_bytes_corrected": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
            "hbm_bytes_per_step": fetch_kb * 1024 * 2 + write_kb * 1024,
            "note": "FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); WRITE_SIZE uncalibrated"}


if __name__ == "__main__":
    n = sum(1 for o in build_plan(synth_state_dict("original", 5, seed=0), "original", 5).ops if o.kind in (2, 8))   # conv launches / step
    out = reduce(sys.argv[1], sys.argv[2], n)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(out)

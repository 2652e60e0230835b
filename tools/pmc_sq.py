#!/usr/bin/env python
"""MFMA utilisation of the conv kernel from a rocprofv3 PMC pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE) of
`HVN_SPLIT=1 HVN_LANES=0 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline`, summed over the
hvn_conv_igemm_f32 dispatches of the LAST plan execution.
mfma_busy_frac = MFMA busy cycles / (GUI-active cycles x 1024 SIMDs); clock = GUI-active cycles / kernel duration.
usage: python tools/pmc_sq.py <pmc.db> <out.json>"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd.plan import build_plan  # noqa: E402
from hover_net_amd.synth import synth_state_dict  # noqa: E402

n = sum(1 for o in build_plan(synth_state_dict("original", 5, seed=0), "original", 5).ops if o.kind in (2, 8))
c = sqlite3.connect(sys.argv[1])
ids = [r[0] for r in c.execute("select dispatch_id, min(start) from counters_collection where (kernel_name like '%igemm%' or kernel_name like '%conv_chain%' or kernel_name like '%dense_grouped%') "
                               "group by dispatch_id order by min(start)")][-n:]
q = ",".join(str(i) for i in ids)
tot = dict(c.execute("select counter_name, sum(value) from counters_collection where dispatch_id in (%s) group by counter_name" % q))
dur = c.execute("select sum(e - s) from (select dispatch_id, min(start) s, max(end) e from counters_collection "
                "where dispatch_id in (%s) group by dispatch_id)" % q).fetchone()[0]
gui = tot.get("GRBM_GUI_ACTIVE", 0.0)
if dur and gui / dur > 4.0:      # the view summed the 8 per-XCD GRBM instances (8 x ~2.3 GHz)
    gui /= 8.0
out = {"launches": len(ids), "sum_duration_ms": dur / 1e6, "counters": tot,
       "clock_GHz": gui / dur if dur else None,
       "mfma_busy_frac": tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024) if gui else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out)

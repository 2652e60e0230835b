#!/usr/bin/env python
"""Where the conv kernels' wave-cycles go: SQ counters of rocprofv3 --pmc passes over `python bench.py --pmc-child` (ONE untimed step on the
single-stream schedule), summed per kernel family over the conv / chain / grouped dispatches of the LAST plan execution of each pass.
Each pass holds at most eight SQ counters, so several databases are merged (a counter present in more than one pass is taken from the first).
Derived ratios (per family):  wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of wave-cycles spent waiting on anything),
wait_inst_lds = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES, valu_active = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES (x4: four SIMDs),
lds_bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (conflict cycles per active LDS cycle), mfma_coexec = MFMA cycles that had
another VALU instruction executing beside them / MFMA busy cycles.
usage: python tools/pmc_stall.py <out.json> <pmc1.db> [<pmc2.db> ...]"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd.plan import build_plan  # noqa: E402
from hover_net_amd.synth import synth_state_dict  # noqa: E402

FAMILIES = (("hvn_conv_igemm_x3g", "x3g (LDS-DMA forms)"), ("hvn_conv_igemm_x3", "x3 (staged forms)"), ("hvn_conv_chain_x3r", "chain x3r"),
            ("hvn_conv_chain", "chain (staged)"), ("hvn_dense_grouped", "dense grouped 5x5"), ("hvn_conv_igemm_f32", "fp32 pipe"))


def family(name):
    for pat, fam in FAMILIES:
        if pat in name:
            return fam
    return "other"


def main():
    out_path, dbs = sys.argv[1], sys.argv[2:]
    n = sum(1 for o in build_plan(synth_state_dict("original", 5, seed=0), "original", 5).ops if o.kind in (2, 8))
    tot = {}
    for db in dbs:
        c = sqlite3.connect(db)
        ids = [r[0] for r in c.execute("select dispatch_id, min(start) from counters_collection where (kernel_name like '%igemm%' or kernel_name like '%conv_chain%' "
                                       "or kernel_name like '%dense_grouped%') group by dispatch_id order by min(start)")][-n:]
        q = ",".join(str(i) for i in ids)
        for name, counter, value in c.execute("select kernel_name, counter_name, sum(value) from counters_collection where dispatch_id in (%s) "
                                              "group by kernel_name, counter_name" % q):
            fam = tot.setdefault(family(name), {})
            key = (db, counter)
            if any(k[1] == counter and k[0] != db for k in fam.get("_src", set())):
                continue
            fam.setdefault("_src", set()).add(key)
            fam[counter] = fam.get(counter, 0.0) + float(value)
    res = {}
    for fam, d in tot.items():
        d.pop("_src", None)
        r = dict(d)
        wc, busy = d.get("SQ_WAVE_CYCLES"), d.get("SQ_BUSY_CYCLES")
        if wc:
            for k, nme in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_WAIT_INST_LDS", "wait_inst_lds")):
                if k in d:
                    r[nme + "_per_wave_cycle"] = d[k] / wc
        if busy:
            for k, nme in (("SQ_ACTIVE_INST_VALU", "valu_active"), ("SQ_ACTIVE_INST_LDS", "lds_active"), ("SQ_ACTIVE_INST_VMEM", "vmem_active"),
                           ("SQ_ACTIVE_INST_ANY", "any_active")):
                if k in d:
                    r[nme + "_per_busy_cycle"] = d[k] / busy
        if d.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_bank_conflict_per_active_cycle"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            r["mfma_coexec_per_mfma_busy"] = d.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / d["SQ_VALU_MFMA_BUSY_CYCLES"]
        if d.get("SQ_INSTS_MFMA"):
            r["valu_insts_per_mfma_inst"] = d.get("SQ_INSTS_VALU", 0.0) / d["SQ_INSTS_MFMA"] if "SQ_INSTS_VALU" in d else None
        res[fam] = r
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    for fam, r in sorted(res.items()):
        print(fam, {k: round(v, 4) for k, v in r.items() if isinstance(v, float) and k[0].islower()})


if __name__ == "__main__":
    main()

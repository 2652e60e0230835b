#!/usr/bin/env python
"""Reads the per-workgroup timestamp dump of ONE conv launch (HVN_CONV_TRACE=<file> python tools/conv_bench.py ...): how long a
workgroup's k-loop and epilogue take, and whether the two workgroups that share a CU run their epilogues at the same time
(lock-step) or under each other's k-loop.  Cycle counter = s_memtime (constant 100 MHz on gfx9: reported in ticks)."""
import sys

import numpy as np

FINE = "--fine" in sys.argv          # dump of the `trace` build (lib.VARIANTS): 8 words per workgroup, + the end of each epilogue phase
a = np.fromfile(sys.argv[1], np.uint64).reshape(-1, 8 if FINE else 4)
a = a[a[:, 2] > 0]
if FINE:
    ph = a[:, [1, 4, 5, 6, 7, 2]].astype(np.int64)      # k-loop end, LDS tile + barrier, residual arrived, values done, stores issued, stores acknowledged
    d = np.diff(ph, axis=1)
    names = ["accumulators -> LDS + barrier", "residual loads return", "values finished", "stores issued", "stores acknowledged"]
    print("epilogue phases (ticks: median / p10 / p90):")
    for i, nm in enumerate(names):
        print("  %-32s %8d %8d %8d" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
t0, t1, t2, hw = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].astype(np.int64), a[:, 3]
hwid = (hw & 0xffffffff).astype(np.int64)
lds = (hw >> np.uint64(32)).astype(np.int64)
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 0x7
xcc = (hwid >> 20) & 0xf            # gfx94x/95x: XCC_ID lives in a different register; kept 0 if absent
loc = ((xcc * 8 + se) * 2 + sh) * 16 + cu
base = t0.min()
k, e = (t1 - t0), (t2 - t1)
print("workgroups %d, span %d ticks; k-loop ticks median %d (p10 %d, p90 %d); epilogue median %d (p10 %d, p90 %d); epilogue/k-loop %.2f"
      % (len(a), t2.max() - base, np.median(k), np.percentile(k, 10), np.percentile(k, 90), np.median(e), np.percentile(e, 10), np.percentile(e, 90),
         np.median(e) / max(1, np.median(k))))
print("distinct (xcc,se,sh,cu) locations seen: %d; LDS bases seen: %s" % (len(np.unique(loc)), np.unique(lds)[:8]))
# overlap of epilogue intervals between workgroups on the same location (the hardware ids do not cover the XCD, so 8 CUs alias:
# restrict to workgroups with blockIdx % 8 equal, i.e. one XCD)
tot_e, both = 0, 0
for x in range(8):
    sel = np.arange(len(a)) % 8 == x
    for l in np.unique(loc[sel]):
        m = sel & (loc == l)
        iv = sorted(zip(t1[m], t2[m]))
        ev = []
        for s_, f_ in iv:
            ev.append((s_, 1))
            ev.append((f_, -1))
        ev.sort()
        depth, last = 0, None
        for t, d in ev:
            if last is not None and depth >= 1:
                tot_e += (t - last) * 1
                if depth >= 2:
                    both += (t - last)
            depth += d
            last = t
print("time with >= 1 epilogue running on a CU: %d ticks; of which BOTH resident workgroups in their epilogue: %.1f %%" % (tot_e, 100.0 * both / max(1, tot_e)))

#!/usr/bin/env python
"""Times ONE conv launch shape through the real library (MiniPlan + Engine, like tests/gpu_util.py) -- the tool behind the
per-layer ablations of profiles/r02_experiments.md.  HVN_CONV_ABLATE selects the ablated instantiation of the 128x128 kernel:
5 = baseline (same instantiation, nothing removed), 1 = no global loads in the k-loop, 2 = no LDS staging either, 3 = MFMAs only,
4 = full k-loop but an epilogue without global traffic.
usage: python tools/conv_bench.py N H CIN COUT K [res] [pre] [x2=CIN2]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from gpu_util import MiniPlan  # noqa: E402
from hover_net_amd import plan as PL  # noqa: E402
from hover_net_amd.engine import Engine  # noqa: E402

n, h, cin, cout, k = (int(v) for v in sys.argv[1:6])
flags = sys.argv[6:]
rng = np.random.default_rng(0)
P = MiniPlan()
pad = (k // 2, k // 2) if k > 1 else (0, 0)
xb = P.buf("x", h, h, cin)
yb = P.buf("y", h, h, cout)
kw = {}
if "res" in flags:
    kw["res"] = PL.View(yb)
if "pre" in flags:
    kw["pre"] = (rng.uniform(0.5, 1.5, cin), rng.normal(0, 0.3, cin))
wt = rng.normal(0, np.sqrt(2.0 / (cin * k * k)), (cout, cin, k, k))
op = P.conv("case", PL.View(xb), PL.View(yb), wt, pad=pad, bn=(rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.2, cout)), relu=1, **kw)
P.pack()
eng = Engine(P, max_batch=n)
eng.arena.normal_()
for _ in range(3):
    eng.run_raw(n)
torch.cuda.synchronize()
reps = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    eng.run_raw(n)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * n * h * h * cout * cin * k * k
print("abl=%s N=%d %dx%d %d->%d k%d %s tile_n=%d: %.3f ms  %.1f TFLOP/s" % (os.environ.get("HVN_CONV_ABLATE", "-"), n, h, h, cin, cout, k, " ".join(flags),
                                                                         eng.ops[0].tile_n, ms, fl / ms / 1e9))

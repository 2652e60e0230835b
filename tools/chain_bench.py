#!/usr/bin/env python
"""One OP_CHAIN launch (or the two CONV launches it replaces, --unfused) at a real shape, timed with HIP events; with
HVN_CHAIN_TRACE=<file> also the per-workgroup phase stamps of the chain kernel (csrc/hvn_conv_chain.hip CH_STAMP).
    python tools/chain_bench.py 32 264 64 256 64 [--x2 64 1] [--post] [--no-res] [--unfused]
args: batch, H = W, K1, C, N2."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from gpu_util import MiniPlan, rand_conv_weight  # noqa: E402
from hover_net_amd import plan as PL  # noqa: E402
from hover_net_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("shape", type=int, nargs=5)
ap.add_argument("--x2", type=int, nargs=2)
ap.add_argument("--post", action="store_true")
ap.add_argument("--no-res", action="store_true")
ap.add_argument("--unfused", action="store_true")
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
n, hw, k1, c, n2 = a.shape
rng = np.random.default_rng(0)
P = MiniPlan()
t2 = PL.View(P.buf("t2", hw, hw, k1))
acc = PL.View(P.buf("acc", hw, hw, c))
out = PL.View(P.buf("out", hw, hw, c)) if a.post else acc
t1 = PL.View(P.buf("t1", hw, hw, n2))
kw = {}
if a.x2:
    kw.update(x2=PL.View(P.buf("bin", (hw - 1) * a.x2[1] + 1, (hw - 1) * a.x2[1] + 1, a.x2[0])), wt2=rand_conv_weight(rng, c, a.x2[0], 1), stride2=a.x2[1])
if not a.no_res and not a.x2:
    kw["res"] = acc
if a.post:
    kw["post"] = (rng.uniform(0.5, 1.5, c), rng.normal(0, 0.3, c))
P.conv("u.conv3", t2, out, rand_conv_weight(rng, c, k1, 1), **kw)
P.conv("v.conv1", out, t1, rand_conv_weight(rng, n2, c, 1), bn=(rng.uniform(0.5, 1.5, n2), rng.normal(0, 0.2, n2)), relu=1,
       pre=None if a.post else (rng.uniform(0.5, 1.5, c), rng.normal(0, 0.3, c)))
if not a.unfused:
    P.fuse_chains()
P.pack()
eng = Engine(P, max_batch=n)
eng.arena.normal_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for r in range(a.reps + 2):
    e0.record()
    eng.run_raw(n)
    e1.record()
    e1.synchronize()
    if r >= 2:
        ts.append(e0.elapsed_time(e1))
fl = sum(o.flops() for o in P.ops) * n
by = 4.0 * n * hw * hw * (k1 + 2 * c + n2 + (0 if (a.no_res or a.x2) else c))
t = float(np.median(ts))
print("%s shape=%s launches=%d  %.1f us  %.1f TFLOP/s  %.2f TB/s (compulsory bytes)" % ("unfused" if a.unfused else "chain", a.shape, len(P.ops), t * 1e3, fl / t / 1e9, by / t / 1e9))
path = os.environ.get("HVN_CHAIN_TRACE")
if path and os.path.exists(path) and not a.unfused:
    d = np.fromfile(path, np.uint64).reshape(-1, 10).astype(np.int64)
    names = ["start", "chunk top", "stage in LDS (barrier)", "k-loop done", "tile written (2 barriers)", "epilogue: y stores issued", "barrier", "GEMM2 done", "all chunks done", "t1' stored"]
    print("workgroups %d; median / p10 / p90 cycles between stamps (one steady-state chunk):" % len(d))
    for i in (2, 3, 4, 5, 6, 7):
        dd = d[:, i] - d[:, i - 1]
        print("  %-28s -> %-28s %8.0f %8.0f %8.0f" % (names[i - 1], names[i], np.median(dd), np.percentile(dd, 10), np.percentile(dd, 90)))
    dd = d[:, 7] - d[:, 1]
    print("  one chunk                                                  %8.0f %8.0f %8.0f" % (np.median(dd), np.percentile(dd, 10), np.percentile(dd, 90)))
    dd = d[:, 9] - d[:, 0]
    print("  workgroup lifetime                                         %8.0f %8.0f %8.0f   (epilogue 2: %.0f)" % (np.median(dd), np.percentile(dd, 10), np.percentile(dd, 90), np.median(d[:, 9] - d[:, 8])))
    span = d[:, 9].max() - d[:, 0].min()
    print("  launch span %d cycles" % span)

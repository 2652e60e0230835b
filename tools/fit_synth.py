#!/usr/bin/env python
"""Exploration / evidence for the bf16 tolerance (cfg 3): fit a trained-like 'fast' network on painted tiles with the repository's
trainer (tests/fit_util.py), then segment held-out tiles with the fp32 and the bf16 network and print the panoptic quality of
bf16 against fp32 (reference metric: metrics/stats_utils.py:178 get_fast_pq, restated in tests/pq_util.py) and against the truth."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import fit_util  # noqa: E402
from pq_util import pq  # noqa: E402
from hover_net_amd import post_proc, run_desc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 240
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
t0 = time.perf_counter()
net, curve = fit_util.fit("fast", None, steps=steps, lr=lr, log=print)
print("fit: %d steps in %.1f s, loss %.3f -> %.3f" % (steps, time.perf_counter() - t0, curve[0], curve[-1]))
imgs, anns = fit_util.painted_tiles(16, 256, seed=999)
o = (256 - 164) // 2
truth = anns[:, o:o + 164, o:o + 164]
tiles = torch.from_numpy(imgs).cuda()
out = {}
for dt in ("fp32", "bf16"):
    net.compute_dtype = dt
    pred = run_desc.infer_step_device(tiles, net).clone()
    inst, _, _ = post_proc.process_batch_device(pred, None, False)
    out[dt] = (pred.cpu().numpy(), inst.cpu().numpy())
pm32, i32 = out["fp32"]
pm16, i16 = out["bf16"]
print("max |p16 - p32| %.4f, max |hv16 - hv32| %.4f" % (np.abs(pm16[..., 0] - pm32[..., 0]).max(), np.abs(pm16[..., 1:] - pm32[..., 1:]).max()))
q = [pq(i32[k], i16[k]) for k in range(len(i32))]
qt = [pq(truth[k], i32[k]) for k in range(len(i32))]
print("instances fp32 %s" % [int(len(np.unique(i)) - 1) for i in i32])
print("instances true %s" % [int(len(np.unique(i)) - 1) for i in truth])
print("PQ bf16 vs fp32: mean %.4f min %.4f" % (np.mean(q), np.min(q)))
print("PQ fp32 vs truth: mean %.4f min %.4f" % (np.mean(qt), np.min(qt)))

#!/usr/bin/env python
"""Round-5 advisor (medium): the weight gradients run on the bf16 pipe with SIX of the nine partial products by default, validated per launch
only.  This is the end-to-end A/B it asked for, made meaningful by the deterministic step (one seed = one checkpoint per arithmetic): the
240-step fit of tests/fit_util.py with the weight gradients on the fp32 pipe | bf16x3 with 9 | with 6 partial products (forward / data-gradient
convs on their default, 6, in all three), two seeds each: loss curve windows, panoptic quality of the fitted network against the painted truth
on 48 held-out tiles, and the bf16-vs-fp32 instance agreement of tools/bf16_pq_table.py.
usage: python tools/wgrad_terms_ab.py > gpurun_out/r06_wgrad_terms_ab.txt"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tools"))
from pq_util import pq  # noqa: E402
from bf16_pq_table import pairs, segment  # noqa: E402
from hover_net_amd import synth_fit  # noqa: E402

imgs, anns = synth_fit.painted_tiles(48, 256, seed=999)
o = (256 - 164) // 2
truth = anns[:, o:o + 164, o:o + 164]
tiles = torch.from_numpy(imgs).cuda()
for wg in ("0", "9", "6"):
    for seed in (0, 1):
        os.environ["HVN_TRAIN_WGRAD_X3"] = wg
        net, curve = synth_fit.fit("fast", None, steps=240, lr=1e-3, seed=seed)
        pm32, i32 = segment(net, tiles, "fp32")
        pm16, i16 = segment(net, tiles, "bf16")
        qt = [pq(truth[k], i32[k]) for k in range(48)]
        tp = n = 0
        for k in range(48):
            la, lb, t = pairs(i32[k], i16[k])
            tp += t
            n += len(la) + len(lb) + 2 * t
        print("weight gradients %-9s seed %d: loss first10 %.3f  steps 100-130 %.4f  last30 %.4f | PQ vs truth %.4f | bf16-vs-fp32 instance agreement %.4f (%d instances)"
              % ({"0": "fp32 pipe", "9": "x3 9 terms", "6": "x3 6 terms"}[wg], seed, np.mean(curve[:10]), np.mean(curve[100:130]), np.mean(curve[-30:]), np.mean(qt),
                 2.0 * tp / max(1, n), n // 2), flush=True)
        net._train_engine = None
        del net
        torch.cuda.empty_cache()

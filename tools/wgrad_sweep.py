#!/usr/bin/env python
"""Weight-gradient split sweep on one MI355X: the K split of `hvn_conv_wgrad_f32` (csrc/hvn_train.hip: launch_wgrad) is set by
two numbers -- the workgroups a launch aims at and the fewest reduction rows a workgroup takes (each workgroup ends with one fp32
atomic per output element of its tile, so short splits trade matrix time for atomic traffic).  Times loss+backward of phase 0 (freeze,
batch 16) and phase 1 (all layers, batch 4) of opt.py:23-142 for a grid of (HVN_WGRAD_WGS, HVN_WGRAD_MIN_ROWS); one process, the
knobs are read per launch.  usage: python tools/wgrad_sweep.py [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd import net_desc  # noqa: E402
from hover_net_amd.synth import synth_state_dict, synth_train_batch  # noqa: E402
from hover_net_amd.train_engine import TrainEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    grid = [(w, r) for w in (512, 1024, 1536, 3072) for r in (256, 512, 1024, 2048)]
    for phase, (freeze, bs) in enumerate(((True, 16), (False, 4))):
        net = net_desc.create_model(mode="original", nr_types=5, input_ch=3, freeze=freeze)
        net.load_state_dict(synth_state_dict("original", 5, seed=0), strict=True)
        net = net.to("cuda")
        eng = TrainEngine(net, bs)
        eng.load_batch(synth_train_batch(bs, "original", 5, seed=1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        print("phase %d (batch %d, freeze %s): loss+backward ms, median of %d" % (phase, bs, freeze, args.reps))
        for wgs, rows in [(None, None)] + grid:
            for k, v in (("HVN_WGRAD_WGS", wgs), ("HVN_WGRAD_MIN_ROWS", rows)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = str(v)
            ts = []
            for i in range(args.reps + 2):
                eng.forward()
                e0.record()
                eng.loss_and_backward()
                e1.record()
                e1.synchronize()
                if i >= 2:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            print("  wgs %-7s min_rows %-7s  %.2f   (min %.2f)" % (wgs or "default", rows or "default", ts[len(ts) // 2], ts[0]), flush=True)
        del eng, net
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

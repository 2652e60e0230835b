#!/usr/bin/env python
"""Training-step timing on one MI355X (BASELINE cfg 5 shapes, single GPU): phase 0 (freeze, batch 16) and phase 1
(all layers, batch 4) of opt.py:23-142, CoNSeP 'original' mode with 5 types, synthetic batch, FusedAdam.
Prints one JSON line per phase: ms per step split into forward / loss+backward / optimizer, steps/s, and the
direct-convolution (algorithmic) conv FLOPs per step (forward + data-gradient + weight-gradient; the 5x5 convs execute 6x fewer as Winograd) over the step time.
usage: python tools/train_bench.py [--steps 10] [--warmup 3] [--phase 0|1|both]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd import net_desc  # noqa: E402
from hover_net_amd.optim import FusedAdam  # noqa: E402
from hover_net_amd.synth import synth_state_dict, synth_train_batch  # noqa: E402
from hover_net_amd.train_engine import TrainEngine  # noqa: E402


def conv_flops(plan, n):
    f = 0.0
    for op in plan.fwd:
        if op.kind == "conv":
            f += 2.0 * n * op.y.h * op.y.w * op.y.c * (op.x.c // op.groups) * op.kh * op.kw
    b = 0.0
    for op in plan.bwd:
        if op.kind == "wgrad":
            b += 2.0 * n * op.dy.h * op.dy.w * op.dy.c * (op.x.c // op.groups) * op.kh * op.kw
        elif op.kind == "dgrad":
            b += 2.0 * n * op.dx.h * op.dx.w * op.dx.c * op.dy.c * op.kh * op.kw      # as executed (dense, dilated)
    return f, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--phase", default="both")
    ap.add_argument("--mode", default="original")
    ap.add_argument("--nr-types", type=int, default=5)
    args = ap.parse_args()
    nt = args.nr_types if args.nr_types > 0 else None
    for phase, (freeze, bs) in enumerate(((True, 16), (False, 4))):
        if args.phase not in ("both", str(phase)):
            continue
        net = net_desc.create_model(mode=args.mode, nr_types=nt, input_ch=3, freeze=freeze)
        net.load_state_dict(synth_state_dict(args.mode, nt, seed=0), strict=True)
        net = net.to("cuda")
        eng = TrainEngine(net, bs)
        opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
        eng.load_batch(synth_train_batch(bs, args.mode, nt, seed=1))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        tf = tb = to = 0.0
        for i in range(args.warmup + args.steps):
            ev[0].record()
            eng.forward()
            ev[1].record()
            eng.loss_and_backward()
            ev[2].record()
            opt.step()
            ev[3].record()
            torch.cuda.synchronize()
            if i >= args.warmup:
                tf += ev[0].elapsed_time(ev[1])
                tb += ev[1].elapsed_time(ev[2])
                to += ev[2].elapsed_time(ev[3])
        k = args.steps
        ff, fb = conv_flops(eng.plan, bs)
        ms = (tf + tb + to) / k
        print(json.dumps({"phase": phase, "freeze": freeze, "batch": bs, "ms_per_step": ms, "forward_ms": tf / k, "loss_backward_ms": tb / k,
                          "optimizer_ms": to / k, "steps_per_s": 1000.0 / ms, "tiles_per_s": bs * 1000.0 / ms,
                          "conv_gflop_forward": ff / 1e9, "conv_gflop_backward": fb / 1e9, "conv_tflops": (ff + fb) / ms / 1e9,
                          "loss": eng.loss_terms()["overall_loss"], "arena_gb": eng.arena.numel() * 4 / 1e9, "grad_gb": eng.gmem.numel() * 4 / 1e9}))
        del eng, net, opt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

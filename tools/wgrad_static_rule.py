#!/usr/bin/env python
"""Round 6: which split of the pixel sum should a DETERMINISTIC weight-gradient launch take?  In deterministic mode the split cannot be a timed
choice (a summation order that depends on the box and the run), so it has to be a function of the launch shape.  This tool times every
distinct weight-gradient launch of the training plans (phase 0: batch 16 frozen encoder; phase 1: batch 4; the fit's 'fast' batch 8) through
hvn_run_train_plan_ws for each target in WGRAD_TARGETS and prints shape, tiles, rows and the times -- the data the static rule in
train_engine.static_wgrad_target is read off.  usage: python tools/wgrad_static_rule.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd import lib as L  # noqa: E402
from hover_net_amd import net_desc  # noqa: E402
from hover_net_amd.synth import synth_state_dict, synth_train_batch  # noqa: E402
from hover_net_amd import train_engine as TE  # noqa: E402

TARGETS = (3072, 1536, 1024, 768, 512, 384, 256)


def main():
    lib = L.lib()
    seen = {}
    for tag, mode, nt, freeze, bs in (("phase0_b16", "original", 5, True, 16), ("phase1_b4", "original", 5, False, 4), ("fit_fast_b8", "fast", None, False, 8)):
        net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3, freeze=freeze)
        net.load_state_dict(synth_state_dict(mode, nt, seed=0), strict=True)
        eng = TE.TrainEngine(net.to("cuda"), bs, deterministic=True)
        eng.load_batch(synth_train_batch(bs, mode, nt, seed=1))
        eng.forward()
        eng.loss_and_backward()
        # a workspace big enough for every target
        ws = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tsz, base = ctypes.sizeof(L.hvn_top), ctypes.addressof(eng.bwd_ops)
        total = {t: 0.0 for t in TARGETS}
        total_best = 0.0
        print("== %s" % tag)
        for i in range(len(eng.bwd_ops)):
            t = eng.bwd_ops[i]
            if t.kind != TE.T_WGRAD:
                continue
            key = (t.kh, t.kw, t.stride, t.x.c, t.dy.c, t.dy.h, t.dy.w, t.groups, int(t.nbatch), int(t._pad))
            x3 = int(t._pad) and t.groups <= 1 and t.dy.c >= 128 and t.x.c % 128 == 0
            bm = 128 if (x3 or t.dy.c >= 128) else (64 if t.dy.c >= 64 else 32)
            bn = 128 if (x3 or t.x.c % 128 == 0) else (64 if t.x.c % 64 == 0 else 32)
            tiles = -(-t.dy.c // bm) * (t.x.c // bn) * t.kh * t.kw * max(1, int(t.nbatch))
            rows = bs * t.dy.h * t.dy.w
            if (tag, key) not in seen:
                ms = {}
                for want in TARGETS:
                    t.mode = want
                    best = 1e9
                    for r in range(4):
                        e0.record()
                        rc = lib.hvn_run_train_plan_ws(base + i * tsz, 1, bs, eng._stream(), ws.data_ptr(), 4 * ws.numel())
                        assert rc == 0, lib.hvn_train_last_error()
                        e1.record()
                        e1.synchronize()
                        if r:
                            best = min(best, e0.elapsed_time(e1))
                    ms[want] = best
                t.mode = 0
                seen[(tag, key)] = ms
                b = min(ms, key=ms.get)
                print("k%dx%d s%d cin %4d cout %4d out %3dx%-3d g%d nb%-2d x3=%d tiles %5d rows %7d | " % (key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7], key[8], 1 if x3 else 0, tiles, rows) +
                      " ".join("%d:%.3f" % (w, ms[w]) for w in TARGETS) + " | best %d" % b, flush=True)
            for w in TARGETS:
                total[w] += seen[(tag, key)][w]
            total_best += min(seen[(tag, key)].values())
        print("sum over launches: " + " ".join("%d:%.2f" % (w, total[w]) for w in TARGETS) +
              " | every launch at its best: %.2f" % total_best)
        del eng, net, ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Training-step timing on one MI355X (BASELINE cfg 5 shapes, single GPU): phase 0 (freeze, batch 16) and phase 1
(all layers, batch 4) of opt.py:23-142, CoNSeP 'original' mode with 5 types, synthetic batch, FusedAdam.
Prints one JSON line per phase: ms per step split into forward / loss+backward / optimizer, steps/s, and the MFMA work of the step
counted the way bench.py counts it: `executed_gflop` = what the matrix pipe issues (the 5x5 convs run as Winograd F(4x4,5x5) in all
three passes: 64 products per 4x4 tile instead of 400; data gradients of strided convs as the dense dilated convolution they execute),
`algorithmic_gflop` = the direct-convolution figure (never a roofline number).  `mfma_frac_of_step` = executed / WHOLE step time /
157.3 TFLOP/s is a lower bound of the conv kernels' fraction (the step also holds BatchNorm, losses, Adam, packing); the fraction over
the conv kernels' own time comes from the rocprofv3 kernel summary of the same command (tools/train_roofline.py).
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
from hover_net_amd import train_engine  # noqa: E402
from hover_net_amd.train_engine import TrainEngine  # noqa: E402


PEAK_FP32_MATRIX_TFLOPS = 157.3


def conv_flops(eng, n):
    """(algorithmic forward, algorithmic backward, executed forward, executed backward) FLOPs of one step."""
    plan = eng.plan

    def tiles(h, w):
        return -(-h // 4) * -(-w // 4)

    af = ab = ef = eb = 0.0
    for op in plan.fwd:
        if op.kind == "conv":
            d = 2.0 * n * op.y.h * op.y.w * op.y.c * (op.x.c // op.groups) * op.kh * op.kw
            af += d
            wino = eng._is_wino(plan.convs[op.wkey]) and op.stride == 1 and op.res is None
            ef += 2.0 * n * 64 * tiles(op.y.h, op.y.w) * op.y.c * op.x.c if wino else d
    for op in plan.bwd:
        if op.kind == "wgrad":
            d = 2.0 * n * op.dy.h * op.dy.w * op.dy.c * (op.x.c // op.groups) * op.kh * op.kw
            ab += d
            wino = op.wkey in eng._du_off and op.stride == 1
            eb += 2.0 * n * 64 * tiles(op.dy.h, op.dy.w) * op.dy.c * op.x.c if wino else d
        elif op.kind == "dgrad":
            d = 2.0 * n * op.dx.h * op.dx.w * op.dx.c * op.dy.c * op.kh * op.kw      # as executed (dense, dilated)
            ab += d
            eb += 2.0 * n * 64 * tiles(op.dx.h, op.dx.w) * op.dx.c * op.dy.c if eng._is_wino(plan.convs[op.wkey]) else d
    return af, ab, ef, eb


def measure(phase, steps=10, warmup=3, mode="original", nt=5, device="cuda", seed_sd=None, deterministic=None):
    """One phase of the two-stage schedule (opt.py:23-142: phase 0 = frozen encoder, batch 16; phase 1 = all layers, batch 4) on `device`:
    a dict with ms per step split into forward / loss+backward / optimizer and the step's MFMA work.  With torch.distributed
    initialised (world > 1) the step is the data-parallel one of run_desc.train_step -- SUM all-reduce of the loss partial sums and of
    the gradient slab in two buckets, the decoder's under the encoder's backward pass -- and the dict also holds the time of the
    slab's all-reduce ALONE (no overlap) on this rank."""
    import torch.distributed as dist

    freeze, bs = ((True, 16), (False, 4))[phase]
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3, freeze=freeze)
    net.load_state_dict(synth_state_dict(mode, nt, seed=0) if seed_sd is None else seed_sd, strict=True)
    net = net.to(device)
    eng = TrainEngine(net, bs, deterministic=deterministic)
    opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    rank = dist.get_rank() if world > 1 else 0
    eng.load_batch(synth_train_batch(bs, mode, nt, seed=1 + rank))
    ar = (lambda t, async_op=False: dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)) if world > 1 else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    tf = tb = to = 0.0
    for i in range(warmup + steps):
        ev[0].record()
        eng.forward()
        ev[1].record()
        eng.loss_and_backward(world=world, all_reduce=ar)
        ev[2].record()
        opt.step()
        ev[3].record()
        torch.cuda.synchronize()
        if i >= warmup:
            tf += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
            to += ev[2].elapsed_time(ev[3])
    k = steps
    # summed HIP-event time of the step's CONV launches (forward + data-gradient implicit GEMMs; the weight-gradient kernel, BatchNorm,
    # transforms, losses and Adam are the rest of the step)
    import ctypes

    from hover_net_amd import lib as L
    L.lib().hvn_profile_enable(1)
    eng.forward()
    eng.loss_and_backward(world=world, all_reduce=ar)
    buf = (ctypes.c_double * 8192)()
    n_conv = L.lib().hvn_profile_conv_ms_list(buf, 8192)
    L.lib().hvn_profile_enable(0)
    conv_ms = float(sum(buf[:n_conv]))
    af, ab, ef, eb = conv_flops(eng, bs)
    ms = (tf + tb + to) / k
    slab_mb = eng.gslab.numel() * 4 / 1e6
    out = {"phase": phase, "freeze": freeze, "batch": bs, "ms_per_step": ms, "forward_ms": tf / k, "loss_backward_ms": tb / k,
           "optimizer_ms": to / k, "steps_per_s": 1000.0 / ms, "tiles_per_s": world * bs * 1000.0 / ms,
           "executed_gflop_forward": ef / 1e9, "executed_gflop_backward": eb / 1e9,
           "algorithmic_gflop_forward": af / 1e9, "algorithmic_gflop_backward": ab / 1e9,
           "algorithmic_speedup": (af + ab) / (ef + eb),
           "timed_conv_launches": int(n_conv), "timed_conv_launch_ms": conv_ms, "timed_conv_launch_share_of_step": conv_ms / ms,
           "plan_ops_per_step": len(eng.fwd_ops) + len(eng.bwd_ops), "deterministic_reduce": bool(eng.deterministic),
           "first_writer_stores": bool(eng.first_store), "branch_streams": bool(eng.branch_streams), "wgrad_stream": bool(eng.wgrad_stream),
           "wgrad_stream_timed_ms_on_off": [round(v, 3) for v in getattr(eng, "wgrad_stream_ms", ())],
           "gradient_slab_mb": slab_mb, "world_size": world,
           "loss": eng.loss_terms()["overall_loss"], "arena_gb": eng.arena.numel() * 4 / 1e9, "grad_gb": eng.gmem.numel() * 4 / 1e9}
    if world > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        dist.barrier()
        e0.record()
        for _ in range(3):
            dist.all_reduce(eng.gslab, op=dist.ReduceOp.SUM)
        e1.record()
        torch.cuda.synchronize()
        out["allreduce_slab_alone_ms"] = e0.elapsed_time(e1) / 3
    eng.gmem.zero_()
    del eng, net, opt
    torch.cuda.empty_cache()
    return out


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
        af, ab, ef, eb = conv_flops(eng, bs)
        ms = (tf + tb + to) / k
        slab_mb = eng.gslab.numel() * 4 / 1e6
        print(json.dumps({"phase": phase, "freeze": freeze, "batch": bs, "ms_per_step": ms, "forward_ms": tf / k, "loss_backward_ms": tb / k,
                          "optimizer_ms": to / k, "steps_per_s": 1000.0 / ms, "tiles_per_s": bs * 1000.0 / ms,
                          "executed_gflop_forward": ef / 1e9, "executed_gflop_backward": eb / 1e9,
                          "algorithmic_gflop_forward": af / 1e9, "algorithmic_gflop_backward": ab / 1e9,
                          "algorithmic_speedup": (af + ab) / (ef + eb),
                          "mfma_frac_of_step": (ef + eb) / ms / 1e9 / PEAK_FP32_MATRIX_TFLOPS,
                          "mfma_frac_note": "executed MFMA FLOPs / whole step time / 157.3 TFLOP/s: a LOWER bound of the conv kernels' fraction",
                          "gradient_slab_mb": slab_mb, "plan_ops_per_step": len(eng.fwd_ops) + len(eng.bwd_ops),
                          "first_writer_stores": bool(eng.first_store), "branch_streams": bool(eng.branch_streams), "wgrad_stream": bool(eng.wgrad_stream),
                          "wgrad_stream_timed_ms_on_off": [round(v, 3) for v in getattr(eng, "wgrad_stream_ms", ())],
                          "allreduce": "not measurable on one GPU: N > 1 all-reduces the gradient slab (%.0f MB fp32) in two buckets + 64 doubles of "
                                       "loss partial sums per step; no RCCL run exists for it (one-GPU box)" % slab_mb,
                          "conv_tiles": {"launch_shapes_timed": sum(1 for k in train_engine._TILE_CHOICE if k[0] == bs),
                                         "re_tiled": sum(1 for k, v in train_engine._TILE_CHOICE.items() if k[0] == bs and v[0] in (64, 320) and v[1] > v[2]),
                                         "static_ms": sum(v[1] for k, v in train_engine._TILE_CHOICE.items() if k[0] == bs),
                                         "chosen_ms": sum(min(v[1], v[2]) if v[0] in (64, 320) and v[2] < v[1] else v[1]
                                                          for k, v in train_engine._TILE_CHOICE.items() if k[0] == bs),
                                         "wgrad_targets": {str(w): sum(1 for k, v in train_engine._TILE_CHOICE.items() if k[0] == "wgrad" and k[1] == bs and v[0] == w)
                                                           for w in train_engine.WGRAD_TARGETS},
                                         "wgrad_static_ms": sum(v[1] for k, v in train_engine._TILE_CHOICE.items() if k[0] == "wgrad" and k[1] == bs),
                                         "wgrad_chosen_ms": sum(v[2] if v[0] != train_engine.WGRAD_TARGETS[0] else v[1]
                                                                for k, v in train_engine._TILE_CHOICE.items() if k[0] == "wgrad" and k[1] == bs),
                                         "note": "TrainEngine.autotune_tiles: one timing per distinct launch shape (sums are over shapes, not launches); HVN_TILE_SELECT=0 keeps the static choices"},
                          "loss": eng.loss_terms()["overall_loss"], "arena_gb": eng.arena.numel() * 4 / 1e9, "grad_gb": eng.gmem.numel() * 4 / 1e9}))
        del eng, net, opt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- tiles/sec of the HoVer-Net hot path on MI355X (BASELINE.json metric, cfg 2).

One *step* = one pass of the whole hot path over one batch of synthetic tiles resident in HBM:
32 uint8 270x270 tiles -> HIP network (original mode, 5 types, fp32) -> infer_step epilogue -> on-GPU
instance separation (Sobel / threshold / CC / watershed) + per-instance table OF THE NETWORK'S OWN OUTPUT
-> [N > 1: RCCL gather of instance maps + record tables to rank 0] -> D2H of instance maps, records and
counts into pinned host memory.  The post-processing of step i runs on a side stream under the network of
step i+1 (hover_net_amd/pipeline.py); nothing inside a step waits on the host.

The checkpoint (round 4): a random-init network emits maps without nuclei, so the flow of
/root/reference/infer/tile.py:308-316,361-386 (network maps -> `process`) had nothing to separate.  By default the
checkpoint is now FITTED at bench start, outside every timed region, with the repository's own trainer
(hover_net_amd/synth_fit.py: ~200 steps of run_desc.train_step on painted H&E-like tiles at CoNSeP's nucleus density),
and the timed tiles are painted tiles of the same kind: the step's instance separation runs on what the network
emits.  `--checkpoint random` keeps the seeded random-init weights; `variants.plus_structured_maps` is round 3's step
(an additional resident batch of structured synthetic maps post-processed per step).

Launch schedule (round 4): the timed step runs the engine's default schedule -- the encoder as two sub-batches on two
HIP streams, the decoder branches on three (bit-equal outputs, tests/test_gpu_chain.py) -- and the roofline leg has
its OWN engine of the same checkpoint on ONE launch stream, where every conv launch can be timed alone;
`variants.single_stream_schedule` is the timed step on that engine.

`value` = tiles/s over exactly --steps steps, inputs resident in HBM when the timed region starts, results on the
host (the bench contract).  SURVEY 8d defines the metric from pinned host memory to host results: that rate is
measured over the same K steps right after and reported at the TOP LEVEL as `value_host_to_host` (and under
`variants.host_to_host`); `with_dict` additionally runs the contour tracing + inst_info_dict assembly of the
structured maps on the host, i.e. everything `post_proc.process` returns; `sustained` repeats the timed step for
>= 8 s; `variants.cfg3_fast_b64_bf16` is BASELINE cfg 3 (fast mode, 6 types, batch 64, bf16) with its own
roofline against the bf16 matrix peak.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  Tiles are independent units: every rank
runs the network + instance separation on its own batch (weak scaling, global batch = 32 N) and the results
are gathered to rank 0 INSIDE the timed step (`infer_tile.gather_to_rank0`, the collective north_star names).
`--scaling strong` keeps a fixed set of --strong-tiles tiles and splits it over the ranks instead.

`roofline` is for the dominant kernel -- the bf16x3 convolution (`hvn_conv_igemm_x3` / its LDS-DMA form `hvn_conv_igemm_x3g`: fp32
operands and accumulation, products on the bf16 matrix pipe): achieved = bf16 MFMA FLOPs its launches of one step ISSUE (6 per fp32
multiply-add, after the Winograd transforms) / summed HIP-event duration of those launches, measured on the single-stream engine (same
plan, same kernels), against the 2.5 PFLOP/s dense bf16 peak; `fp32_equivalent_tflops` counts each fp32 product once (comparable with
round 3's `achieved`); `other_launches` = the fp32-pipe launches and the Winograd transforms.  `algorithmic_speedup` =
direct-convolution FLOPs / executed FLOPs (what Winograd removes) is reported beside it, never folded into `frac`.
`variants.train_step` (BASELINE cfg 5 at N = 1: both phases of the two-stage schedule) and `variants.wsi_8k` (cfg 4 scaled to one GPU:
an 8192^2 synthetic slide) are timed outside the headline's timed region; `flood_whole_tile_replays` counts the watershed's exact
whole-tile replays over the network's own output of the timed step.

`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment) starts itself under
`python -m torch.distributed.run --nproc-per-node N`; under the launcher it runs as one rank.  `cpu_baseline` (N = 1 only) times the CPU oracle -- torch fp32
restatement of the network + the C / python restatement of `process()` -- on a bounded sample of the same
tiles on this box's host cores.  The reference itself is python under /root/reference and cannot travel to
the GPU box, so kind = "port"; profiles/ holds the reference's own timing taken in the build container.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_MATRIX_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 MFMA
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # same guide: ~2.5 PFLOP/s dense bf16 (cfg 3 only: --dtype bf16)


def _control_flow_selftest(args, rank, world, dev):
    """The distributed skeleton of main() with the GPU pipeline replaced by correctly shaped dummy tensors (see the flag's help)."""
    import torch
    import torch.distributed as dist

    from hover_net_amd import infer_tile

    b, hw, max_inst = args.batch, 80, 80 * 80 // 13 + 1
    gather = infer_tile.gather_to_rank0 if world > 1 else None

    def submit():
        out = (torch.full((b, hw, hw), rank + 1, dtype=torch.int32), torch.zeros((b, max_inst, 56), dtype=torch.uint8),
               torch.full((b,), rank, dtype=torch.int32))
        return gather(out) if gather is not None else out

    def fence():
        if world > 1:
            dist.barrier()

    def timed(steps):
        fence()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = submit()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    for _ in range(args.warmup):
        submit()
    dt, out = timed(args.steps)
    per_rank = None
    if world > 1:                      # the per-rank diagnosis of main(): every rank's own numbers all_gathered
        mine = torch.tensor([float(rank), 0.0], dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(t[0]) for t in allr]
        assert per_rank == [float(r) for r in range(world)]
    if rank == 0:
        assert out[0].shape[0] == world * b and out[0][::b, 0, 0].tolist() == list(range(1, world + 1)), "gather order = rank order"
        assert out[2][::b].tolist() == list(range(world))
    else:
        assert out is None or world == 1
    reps, t_end = 0, time.perf_counter() + min(args.sustain_seconds, 1.0)
    while True:                                   # the sustained leg's stop vote: every rank leaves in the same iteration
        for _ in range(args.steps):
            submit()
        reps += args.steps
        flag = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0])
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() == 0.0:
            break
    if rank == 0:
        print(json.dumps({"metric": "control-flow self-test (no kernels, nothing measured)", "value": 0.0, "unit": "tiles/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "data": "dummy",
                          "config": {"global_batch": world * b, "world_size": world, "sustained_steps": reps}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: run this very command line as N ranks of one node (one process per GPU) under
    torch.distributed.run, on a free local port; returns its exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.call(cmd, env=env)


def roofline_account(timed, per_ms, batch, dtype, n_prof=5):
    """The `roofline` object from the plan's timed launches (`timed`: the CONV / CHAIN / WINO_IN / WINO_OUT ops in launch order) and their
    per-launch times in ms (`per_ms`, same order).  Pure arithmetic (tests/test_bench_roofline.py feeds it made-up times).

    fp32 engine = two matrix pipes: the launches `Plan.mark_x3` put on csrc/hvn_conv_x3.hip issue bf16 MFMAs (6 | 9 per fp32 product: exact
    three-way bf16 splits of fp32 operands, fp32 accumulation); the rest (d0's chained 1x1 pairs, grouped 5x5, d0's first 1x1) issue fp32
    MFMAs; the Winograd transforms issue none.  The DOMINANT kernel is the bf16x3 one: `achieved` / `peak` / `frac` describe it on ITS
    pipe; `other_launches` the fp32-pipe launches + transforms; `whole_step` both, as ideal matrix time (each launch's executed FLOPs / its
    pipe's peak) over measured time.  Without bf16x3 launches (HVN_X3=0, the bf16 engine): one pipe, round 3's accounting."""
    convs = [o for o in timed if o.kind in (2, 8)]
    algo_flops = sum(o.flops() for o in convs) * batch
    exec_flops = sum(o.extra.get("exec_flops", o.flops()) for o in convs) * batch
    launches, n_conv = len(per_ms), len(convs)
    ms = float(sum(per_ms))
    note = ("achieved = MFMA FLOPs executed by the launches (Winograd-domain GEMMs counted as issued; SURVEY 8d's direct-convolution figure is "
            "`algorithmic_gflop_per_step` / batch) / summed HIP-event time of those launches, per-launch median of %d passes, single stream" % n_prof)
    x3_ms = x3_eq = x3_bf16 = 0.0
    ch_ms = ch_eq = ch_bf16 = 0.0          # round 5: the chained seams on the bf16 pipe (hvn_conv_chain_x3 | _x3r), their own kernels
    rest_ms = rest_flops = 0.0
    n_x3 = n_ch = 0
    if dtype == "fp32" and launches == len(timed):
        for o, t in zip(timed, per_ms):
            fl = o.extra.get("exec_flops", o.flops()) * batch if o.kind in (2, 8) else 0.0
            if o.kind == 2 and o.extra.get("x3"):
                x3_ms += float(t); x3_eq += fl; x3_bf16 += fl * int(o.extra["x3"]); n_x3 += 1
            elif o.kind == 8 and o.extra.get("x3"):
                ch_ms += float(t); ch_eq += fl; ch_bf16 += fl * int(o.extra["x3"]); n_ch += 1
            else:
                rest_ms += float(t); rest_flops += fl
    if n_x3 == 0:
        peak = PEAK_FP32_MATRIX_TFLOPS if dtype == "fp32" else PEAK_BF16_MATRIX_TFLOPS
        achieved = exec_flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "hvn_conv_igemm_f32 (+ hvn_conv_chain_f32)" if dtype == "fp32" else "hvn_conv_igemm_bf16",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                "flops_per_launch": exec_flops / max(1, n_conv), "avg_launch_ms": ms / max(1, launches),
                "timed_launches_per_step": launches, "conv_launches_per_step": n_conv, "conv_ms_per_step": ms,
                "executed_gflop_per_step": exec_flops / 1e9, "algorithmic_gflop_per_step": algo_flops / 1e9,
                "algorithmic_speedup": algo_flops / exec_flops, "note": note + " (incl. the Winograd transform launches)"}
    ach_x3 = x3_bf16 / (x3_ms * 1e-3) / 1e12
    ideal_ms = 1e3 * ((x3_bf16 + ch_bf16) / (PEAK_BF16_MATRIX_TFLOPS * 1e12) + rest_flops / (PEAK_FP32_MATRIX_TFLOPS * 1e12))
    chained = None
    if n_ch:
        chained = {"what": "hvn_conv_chain_x3 / hvn_conv_chain_x3r (same bits, picked per seam by time): d0's residual seams (conv3 + residual -> next conv1 "
                           "in one launch), both GEMMs on the bf16 pipe",
                   "launches_on_register_resident_form": sum(1 for o in timed if o.kind == 8 and o.extra.get("x3") and o.extra.get("tile_form") == 1152),
                   "launches": n_ch, "ms_per_step": ch_ms, "bf16_mfma_gflop_per_step": ch_bf16 / 1e9, "achieved": ch_bf16 / (ch_ms * 1e-3) / 1e12,
                   "peak": PEAK_BF16_MATRIX_TFLOPS, "frac": ch_bf16 / (ch_ms * 1e-3) / 1e12 / PEAK_BF16_MATRIX_TFLOPS,
                   "fp32_equivalent_tflops": ch_eq / (ch_ms * 1e-3) / 1e12,
                   "note": "HBM-bound by construction (4.0 .. 6.3 GB of compulsory bytes per seam at batch 32): see profiles/r05_traffic_by_kernel.txt"}
    return {
        "bound": "mfma", "kernel": "hvn_conv_igemm_x3 / hvn_conv_igemm_x3g (fp32 convolution, products on the bf16 matrix pipe from exact bf16x3 splits of "
                                   "the fp32 operands; the second stages both operands by LDS-DMA -- same bits, picked per launch shape by time)",
        "achieved": ach_x3, "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach_x3 / PEAK_BF16_MATRIX_TFLOPS, "traffic": None,
        "achieved_counts": "bf16 MFMA FLOPs ISSUED by the bf16x3 launches (%d per fp32 multiply-add) -- not comparable with rounds 1-3's `achieved` "
                           "(fp32 MFMA FLOPs): that quantity is `fp32_equivalent_tflops` here and `whole_step.fp32_equivalent_tflops` for all launches"
                           % (int(x3_bf16 / x3_eq + 0.5) if x3_eq else 0),
        "launches_on_lds_dma_form": sum(1 for o in timed if o.kind == 2 and o.extra.get("x3") and o.extra.get("tile_form") in (896, 640)),
        "launches": n_x3, "ms_per_step": x3_ms, "avg_launch_ms": x3_ms / n_x3, "flops_per_launch": x3_bf16 / n_x3,
        "bf16_mfma_gflop_per_step": x3_bf16 / 1e9, "fp32_products_gflop_per_step": x3_eq / 1e9,
        "fp32_equivalent_tflops": x3_eq / (x3_ms * 1e-3) / 1e12,
        # round-5 verdict, next #7: the two figures a reader otherwise recomputes.  6 (9) bf16 MFMA FLOPs are issued per fp32 multiply-add, so
        # the most fp32-equivalent work this scheme can ever deliver is peak / 6 (/ 9): `useful_frac_of_scheme_peak` is against THAT
        # (numerically equal to `frac`: issued / peak == useful / (peak / terms)); `frac_of_bf16_peak_useful` counts each product once
        "scheme_peak_fp32_equivalent_tflops": PEAK_BF16_MATRIX_TFLOPS / (x3_bf16 / x3_eq) if x3_eq else None,
        "useful_frac_of_scheme_peak": (x3_eq / (x3_ms * 1e-3) / 1e12) / (PEAK_BF16_MATRIX_TFLOPS / (x3_bf16 / x3_eq)) if x3_eq else None,
        "frac_of_bf16_peak_useful": x3_eq / (x3_ms * 1e-3) / 1e12 / PEAK_BF16_MATRIX_TFLOPS,
        "chained_seams": chained,
        "other_launches": {"what": "fp32-MFMA launches (hvn_conv_igemm_f32, hvn_dense_grouped*, hvn_conv_chain_f32 when HVN_X3_CHAIN is off) + Winograd transform launches",
                           "launches": launches - n_x3 - n_ch, "ms_per_step": rest_ms, "executed_gflop_per_step": rest_flops / 1e9,
                           "achieved": rest_flops / (rest_ms * 1e-3) / 1e12, "peak": PEAK_FP32_MATRIX_TFLOPS,
                           "frac": rest_flops / (rest_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS},
        "whole_step": {"conv_ms_per_step": ms, "ideal_matrix_ms": ideal_ms, "frac": ideal_ms / ms,
                       "fp32_equivalent_tflops": exec_flops / (ms * 1e-3) / 1e12,
                       "what": "ideal = each launch's executed MFMA FLOPs / the dense peak of the pipe it issues on (bf16 2500, fp32 157.3 TFLOP/s); "
                               "fp32_equivalent = fp32 multiply-adds of the executed GEMMs (each counted once) / time: comparable with round 3's "
                               "`achieved` (102.5), not a fraction of any one pipe's peak"},
        "timed_launches_per_step": launches, "conv_launches_per_step": n_conv, "conv_ms_per_step": ms,
        "executed_gflop_per_step": exec_flops / 1e9, "algorithmic_gflop_per_step": algo_flops / 1e9, "algorithmic_speedup": algo_flops / exec_flops,
        "note": note}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--mode", default="original")
    ap.add_argument("--nr-types", type=int, default=5)
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--strong-tiles", type=int, default=1024, help="size of the fixed tile set of --scaling strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the host_to_host / with_dict / sustained legs")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the variants.cfg3_fast_b64_bf16 leg (BASELINE cfg 3)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two `rocprofv3 --pmc` child runs of this script); the last committed "
                         "profiles/*_pmc_traffic.json is quoted instead, and the line says so")
    ap.add_argument("--traffic-timeout", type=float, default=150.0, help="seconds per PMC child run")
    ap.add_argument("--pmc-child", action="store_true", help="internal: one untimed step and nothing else (the run rocprofv3 --pmc wraps)")
    ap.add_argument("--sustain-seconds", type=float, default=8.0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--checkpoint", default="fitted", choices=("fitted", "random"),
                    help="fitted (default): a checkpoint fitted at bench start with the repo's own trainer on painted tiles, so that the network's "
                         "output holds nuclei and the timed instance separation works on it; random: seeded random-init weights + noise tiles "
                         "(round 3's workload: the step then also post-processes a resident batch of structured maps)")
    ap.add_argument("--fit-steps", type=int, default=200)
    ap.add_argument("--fit-init", default="synth", choices=("synth", "kaiming"))
    ap.add_argument("--quiet-net-output", action="store_true",
                    help="bias the NP head of the random-init checkpoint towards background, so that the network's own output "
                         "holds no nuclei and the instance-separation load of the step comes from the structured maps only")
    ap.add_argument("--control-flow-selftest", action="store_true",
                    help="CI only (tests/test_bench_dist.py): run the N-rank control flow of this script -- barriers, max-over-ranks clock, "
                         "the per-batch gather to rank 0, the sustained leg's stop vote, the closing barrier -- on CPU tensors over gloo with "
                         "a stand-in for the GPU pipeline that moves correctly shaped dummy results.  Measures nothing; the JSON line says so.")
    ap.add_argument("--dtype", default="fp32", choices=("fp32", "bf16"),
                    help="fp32 = the headline configuration (BASELINE cfg 2); bf16 = cfg 3 (use with --mode fast --nr-types 6 --batch 64)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip variants.train_step (BASELINE cfg 5: both phases of the training schedule)")
    ap.add_argument("--no-wsi-leg", action="store_true", help="skip variants.wsi_8k (BASELINE cfg 4 scaled to one GPU)")
    ap.add_argument("--wsi-size", type=int, default=8192)
    ap.add_argument("--wsi-leg", action="store_true", help="run the whole-slide leg even with --no-variants (tests)")
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct resident tile batches per rank the timed loop cycles through (round-5 verdict, next #7: rounds 1-5 timed the SAME 32 "
                         "tiles every step; the instance load then never changes and every cache sees the same addresses)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(_self_launch(args.gpus))       # no launcher around us: become N ranks

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    selftest = args.control_flow_selftest
    if selftest:
        dev = torch.device("cpu")
        args.no_roofline = args.no_cpu_baseline = True
        torch.cuda.synchronize = lambda *a, **k: None       # nothing to wait for: the stand-in pipeline is synchronous
    else:
        # HVN_BENCH_SHARED_GPU=1 (tests/test_gpu_two_ranks_one_gpu.py): every rank on cuda:0, collectives over gloo with device tensors staged
        # through the host -- the driver's exact N-rank command path with the real kernels on a one-GPU box; measures nothing comparable
        local = 0 if os.environ.get("HVN_BENCH_SHARED_GPU", "0") != "0" else local
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    shared_gpu = (not selftest) and os.environ.get("HVN_BENCH_SHARED_GPU", "0") != "0"
    cdev = torch.device("cpu") if (selftest or shared_gpu) else dev          # where the small control collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest or shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if selftest:
        return _control_flow_selftest(args, rank, world, dev)

    from hover_net_amd import infer_tile, post_proc, run_desc
    from hover_net_amd import lib as L
    from hover_net_amd import net_desc
    from hover_net_amd.pipeline import TilePipeline
    from hover_net_amd.synth import synth_pred_maps, synth_state_dict, synth_tiles

    from hover_net_amd import synth_fit

    nt = args.nr_types if args.nr_types > 0 else None
    size = 270 if args.mode == "original" else 256
    fitted = args.checkpoint == "fitted" and not args.pmc_child       # HBM traffic does not depend on the weights' values
    fit_info = {}

    def make_checkpoint(mode, nr_types, sz, seed=0):
        """-> (state_dict on the host, info).  Outside every timed region."""
        if not fitted:
            return synth_state_dict(mode, nr_types, seed=seed), {"kind": "random-init (seeded)"}
        t_fit = time.perf_counter()
        info = None
        if rank == 0:            # ONE fit per job: the other ranks receive rank 0's weights (N x 25 s of identical fitting otherwise)
            tnet, curve = synth_fit.fit(mode, nr_types, steps=args.fit_steps, batch=8, lr=1e-3, seed=seed, init=args.fit_init,
                                        density=synth_fit.consep_density(sz))
            sd_ = {k: v.detach().cpu().clone() for k, v in tnet.state_dict().items()}
            tnet._train_engine = None
            del tnet
            torch.cuda.empty_cache()
            info = {"kind": "fitted at bench start%s: %d steps of run_desc.train_step (batch 8, Adam 1e-3, init %s) on painted tiles"
                            % (" on rank 0 and broadcast" if world > 1 else "", args.fit_steps, args.fit_init),
                    "loss_first10": float(np.mean(curve[:10])), "loss_last10": float(np.mean(curve[-10:]))}
        else:
            sd_ = {k: v.detach().cpu().clone() for k, v in net_desc.create_model(mode=mode, nr_types=nr_types, input_ch=3).state_dict().items()}
        if world > 1:
            for k in sd_:           # same keys in the same order on every rank (the module's own state_dict)
                t = sd_[k].to(cdev).contiguous()
                dist.broadcast(t, 0)
                sd_[k] = t.cpu()
            box = [info]
            dist.broadcast_object_list(box, 0)
            info = box[0]
        info["seconds"] = time.perf_counter() - t_fit
        return sd_, info

    def make_tiles(n, sz, seed):
        if not fitted:
            return synth_tiles(n, sz, seed=seed)
        return synth_fit.painted_tiles(n, sz, seed, *synth_fit.consep_density(sz))[0]

    sd, fit_info = make_checkpoint(args.mode, nt, size)
    if args.quiet_net_output:
        sd["decoder.np.u0.conv.bias"] = torch.tensor([8.0, -8.0])

    def make_net(schedule=None, mode=args.mode, nr_types=nt, sd_=None, batch=args.batch, dtype=args.dtype):
        n_ = net_desc.create_model(mode=mode, nr_types=nr_types, input_ch=3)
        n_.load_state_dict(sd if sd_ is None else sd_, strict=True)
        n_.max_batch = batch
        n_.compute_dtype = dtype
        n_.launch_schedule = schedule
        return n_.to(dev).eval()

    if args.pmc_child:              # counters are collected per dispatch by the wrapping rocprofv3: ONE plan execution on one launch stream
        net = make_net((1, 0))
        run_desc.infer_step_device(torch.from_numpy(make_tiles(args.batch, size, 1)).to(dev), net)
        torch.cuda.synchronize(dev)
        return
    net = make_net()           # the engine's default launch schedule (fp32: two encoder sub-batches + decoder branch streams)
    # the work of one step on this rank: `sub` batches of `args.batch` tiles
    if args.scaling == "strong":
        lo, hi = infer_tile.shard_range(args.strong_tiles, rank, world)
        sub = max(1, -(-(hi - lo) // args.batch))
        tiles_per_step_global = args.strong_tiles
    else:
        sub = 1
        tiles_per_step_global = world * args.batch
    rot = max(1, args.rotate)
    sets_host = [[torch.from_numpy(make_tiles(args.batch, size, 1 + rank + 1000 * j + 100000 * r)).pin_memory() for j in range(sub)] for r in range(rot)]
    sets_dev = [[t.to(dev) for t in hs] for hs in sets_host]       # `rot` distinct sets of this rank's work, all resident in HBM
    tiles_host, tiles = sets_host[0], sets_dev[0]
    turn = [0]

    class _Rotating:
        """The rank's work of one step, a different resident set every step (iterating takes the next set)."""
        def __init__(self, sets):
            self.sets = sets

        def __iter__(self):
            turn[0] += 1
            return iter(self.sets[turn[0] % len(self.sets)])

    rot_dev, rot_host = _Rotating(sets_dev), _Rotating(sets_host)
    # Structured synthetic maps (painted, partly touching elliptical nuclei, hover_net_amd.synth.synth_pred_maps; 2..8 per
    # 80x80, CoNSeP: 3.8): with a FITTED checkpoint the network's own output carries the instance-separation load and these are
    # only `variants.plus_structured_maps` (round 3's step) and the stage split's post-processing sample; with a random-init
    # checkpoint (0 instances in the network output) the step also post-processes them, as in round 3.
    eng0 = net.engine(args.batch)
    out_hw = eng0.plan.geo["out"]
    structured_np = synth_pred_maps(args.batch, out_hw, out_hw, nt, seed=100 + rank, k_lo=2, k_hi=8)[0]
    structured = torch.from_numpy(structured_np).to(dev)

    # network of batch i+1 (main stream) overlaps the post-processing of batch i (side stream)
    pipe = TilePipeline(net, nr_types=nt, return_centroids=True)
    pipe.time_gather = world > 1
    gather = infer_tile.gather_to_rank0 if world > 1 else None
    # does the network's own output hold nuclei?  (one untimed pass; a fit that did not converge falls back to round 3's step)
    probe = pipe.submit(tiles[0], to_host=True)
    pipe.wait()
    net_inst = int(probe[2].sum().item())
    extra = None if (fitted and net_inst > 0) else structured
    if fitted and net_inst == 0:
        fit_info["note"] = "the fitted network emitted 0 instances: the step also post-processes the structured maps (round 3's step)"

    def step(src=rot_dev, extra_maps=extra, p=None):
        out = None
        for t in src:
            out = (p or pipe).submit(t, extra_maps=extra_maps, gather=gather, to_host=True)
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(fn, steps):
        fence()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = fn()
        # this rank's own clock: last result of this rank on the host (its side stream drained), BEFORE the closing barrier
        torch.cuda.synchronize(dev)
        last_rank_dt[0] = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    last_rank_dt = [0.0]
    for _ in range(args.warmup):
        step()
    if world > 1:
        pipe.gather_ms()                 # drop the warm-up's gather timings
    dt, out = timed(step, args.steps)
    n_inst = int(out[2].sum().item()) if (rank == 0 and out is not None) else 0
    # which replay the marker-controlled watershed of the LAST timed step's network output took on this rank (one of the `rot` resident
    # sets): whole-tile replays are the exact one-lane fallback a mixed-label marker tie forces (csrc/hvn_postproc.hip)
    flood = pipe.flood_stats() if extra is None else None
    if rank == 0 and world > 1:
        assert out[0].shape[0] == world * args.batch, "rank 0 must hold every rank's instance maps after the gather"
    per_rank = None
    if world > 1:
        # diagnosis of a bad scaling curve from ONE run: every rank's own step time (its clock stops when its own last result is on
        # the host, before the closing barrier) and the mean duration of its `gather` call on the side stream
        mine = torch.tensor([1e3 * last_rank_dt[0] / args.steps, pipe.gather_ms()], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"step_ms": [float(t[0]) for t in allr], "gather_ms": [float(t[1]) for t in allr],
                    "what": "per rank: wall time per step up to its own last result (before the closing barrier); mean HIP-event time of the "
                            "per-batch gather to rank 0 on the side stream (overlaps the next network pass)"}

    ms_per_step = 1e3 * dt / args.steps
    result = {
        "metric": "tiles/sec (%dx%d, batch %d) end-to-end incl. watershed" % (size, size, args.batch),
        "value": tiles_per_step_global * args.steps / dt,
        "unit": "tiles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic" if not shared_gpu else "synthetic (HVN_BENCH_SHARED_GPU: all ranks on ONE GPU over gloo -- a control-path run, not a measurement)",
        "config": {"workload": "CoNSeP '%s' mode seg+class (NP+HV+NC, %s types), batch %d of %dx%d uint8 %s tiles per GPU resident in HBM, "
                               "%s checkpoint%s: network + infer_step epilogue + on-GPU instance separation and instance table of %s, %s"
                               "D2H of instance maps + records to pinned host memory"
                               % (args.mode, nt, args.batch, size, size, "painted H&E-like (CoNSeP nucleus density)" if fitted else "noise",
                                  "fitted (trained-like, made at bench start outside the timed region)" if fitted else "random-init (seeded)",
                                  ", NP head biased to background" if args.quiet_net_output else "",
                                  "the network's own output" if extra is None else "the network output AND of a resident batch of structured synthetic maps",
                                  "RCCL gather to rank 0, " if world > 1 else ""),
                   "global_batch": tiles_per_step_global, "world_size": world, "batches_per_step_per_rank": sub,
                   "resident_sets_rotated": rot,
                   "instances_last_step": n_inst,
                   "instances_from": "network output" if extra is None else "structured synthetic maps (the network output of this checkpoint: %d instances)" % net_inst,
                   "checkpoint": fit_info,
                   "arithmetic": ("fp32 activations, weights, accumulation and outputs; the MFMA-bound conv launches form their products on the bf16 "
                                  "matrix pipe from exact three-way bf16 splits of the fp32 operands (x = h + m + l, 8 significand bits each; "
                                  "%s partial products per product, fp32 accumulate: csrc/hvn_conv_x3.hip), the others on the fp32 matrix pipe; "
                                  "logits within 1e-3 of the fp32 oracle (tests/test_gpu_trained_like.py, test_gpu_x3.py); variants.fp32_mfma_only = "
                                  "every launch on the fp32 pipe" % os.environ.get("HVN_X3", "6")) if (args.dtype == "fp32" and os.environ.get("HVN_X3", "6") != "0")
                                 else "%s MFMA, fp32 accumulation" % args.dtype,
                   "parallelism": "tile-sharded x%d, %s" % (world, "gather to rank 0 per batch" if world > 1 else "single GPU"),
                   "execution": "network: engine default launch schedule (n_split %d, %d decoder branch streams); post-processing + gather + D2H on a "
                                "side stream under the next network pass" % (eng0.n_split, eng0.n_lane_streams)},
    }
    if per_rank is not None:
        result["config"]["per_rank"] = per_rank
    if flood is not None:
        result["flood_whole_tile_replays"] = flood["whole_tile_replays"]
        result["config"]["flood"] = dict(flood, what="marker-controlled watershed of the network's own output, last timed step, rank 0: mask components "
                                                     "by the replay that flooded them; whole_tile_replays = exact whole-tile fallbacks (a marker tie between labels "
                                                     "that the component replay could not prove harmless)")

    # ---- per-stage split of one batch (rank 0): each stage ALONE on the launch stream, warmed, median of 5 passes ------
    # (a stage alone is not a share of the pipelined step: there the post-processing of batch i runs under the network of
    #  batch i+1; `network` <= `ms_per_step` holds because the step contains one whole network pass)
    if rank == 0:
        def med_ms(fn, reps=5, warm=2, inner=1):
            """median over `reps` of (time of `inner` back-to-back calls) / inner: with inner > 1 the host runs ahead of the GPU as
            it does in the pipelined step (a single pass from an idle stream pays ~0.7 ms of launch gaps over ~180 launches)"""
            ts = []
            for r in range(warm + reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(inner):
                    out_ = fn()
                e1.record()
                e1.synchronize()
                if r >= warm:
                    ts.append(e0.elapsed_time(e1) / inner)
            return sorted(ts)[len(ts) // 2], out_

        torch.cuda.synchronize(dev)
        net_ms, pred = med_ms(lambda: run_desc.infer_step_device(tiles[0], net), reps=7, warm=2, inner=max(4, args.steps // 2))
        pred = pred.clone()
        ppn_ms, (inst_n, _r, counts_n) = med_ms(lambda: post_proc.process_batch_device(pred, nr_types=nt, return_centroids=True))
        pp_ms, (inst, rec, counts) = med_ms(lambda: post_proc.process_batch_device(structured, nr_types=nt, return_centroids=True))
        d2h_ms, host = med_ms(lambda: [t.cpu() for t in (inst, rec, counts)])
        t1 = time.perf_counter()
        rec_h, inst_h = host[1].numpy(), host[0].numpy()
        dicts = [post_proc.records_to_dict(rec_h[i].view(post_proc._REC_DTYPE).reshape(-1), nt, inst_h[i]) for i in range(inst_h.shape[0])]
        dict_ms = 1e3 * (time.perf_counter() - t1)
        result["config"]["stage_ms"] = {"network": net_ms, "postproc_network_output": ppn_ms, "instances_in_network_output": int(counts_n.sum().item()),
                                        "postproc_structured": pp_ms, "d2h_results": d2h_ms,
                                        "host_contours_and_dict": dict_ms, "instances_in_dicts": sum(len(d) for d in dicts),
                                        "how": "each stage alone on the launch stream, median of 5 warmed measurements (network: median of 7 measurements of %d back-to-back "
                                               "passes on the timed step's launch schedule)" % max(4, args.steps // 2)}

    # ---- variants (untimed by the driver; same K steps each) ---------------------------------------------------------
    net_rf = None              # the single-stream engine of the same checkpoint (roofline leg, single_stream_schedule variant)
    if not args.no_variants:
        variants = {}
        for _ in range(2):
            step(rot_host)            # (the pipeline's two upload slots and its copy stream are made at first use: rounds 3-5 timed that inside this leg, ~2 ms per step over 20 steps)
        dt_h, _ = timed(lambda: step(rot_host), args.steps)
        variants["host_to_host"] = {"value": tiles_per_step_global * args.steps / dt_h, "unit": "tiles/s", "ms_per_step": 1e3 * dt_h / args.steps,
                                    "what": "SURVEY 8d's definition: tiles start in pinned host memory (H2D inside the step), results end in pinned host memory"}
        result["value_host_to_host"] = variants["host_to_host"]["value"]
        if world == 1:
            prev = [None]

            def step_dict():
                o = step(rot_host)
                done = prev[0]
                prev[0] = (o, torch.cuda.Event())
                prev[0][1].record(pipe.side)
                if done is not None:                # host half of process() for the PREVIOUS step, under this step's GPU work
                    done[1].synchronize()
                    ih, rh = done[0][0].numpy(), done[0][1].numpy()
                    for i in range(ih.shape[0]):
                        post_proc.records_to_dict(rh[i].view(post_proc._REC_DTYPE).reshape(-1), nt, ih[i])
                return o

            def steps_dict_then_flush():
                # K GPU steps carry K-1 host halves (each under the NEXT step's GPU work); the last step's host half runs here,
                # inside the timed region, so that exactly K dictionaries sets are built per K steps
                calls[0] += 1
                o = step_dict()
                if calls[0] % args.steps == 0 and prev[0] is not None:
                    done, prev[0] = prev[0], None
                    done[1].synchronize()
                    ih, rh = done[0][0].numpy().copy(), done[0][1].numpy().copy()
                    for i in range(ih.shape[0]):
                        post_proc.records_to_dict(rh[i].view(post_proc._REC_DTYPE).reshape(-1), nt, ih[i])
                return o

            calls = [0]
            dt_d, _ = timed(steps_dict_then_flush, args.steps)
            variants["with_dict"] = {"value": tiles_per_step_global * args.steps / dt_d, "unit": "tiles/s",
                                     "what": "host_to_host + contour tracing and inst_info_dict assembly on the host (one thread), "
                                             "overlapped with the next step's GPU work"}
        if world == 1:
            # the same step on ONE launch stream (the roofline leg's engine; same plan and kernels, bit-equal outputs:
            # tests/test_gpu_chain.py::test_network_on_sub_batch_and_branch_streams_is_bit_equal)
            net_rf = make_net((1, 0))
            pipe_rf = TilePipeline(net_rf, nr_types=nt, return_centroids=True)
            for _ in range(args.warmup):
                step(p=pipe_rf)
            dt2, _ = timed(lambda: step(p=pipe_rf), args.steps)
            variants["single_stream_schedule"] = {"value": tiles_per_step_global * args.steps / dt2, "unit": "tiles/s", "ms_per_step": 1e3 * dt2 / args.steps,
                                                  "what": "the timed step with the network on ONE HIP stream (n_split 1, no branch streams): round 3's headline "
                                                          "schedule, the one the roofline leg times its launches on"}
            del pipe_rf
            if args.dtype == "fp32" and os.environ.get("HVN_X3", "6") != "0":
                # the same step with EVERY conv launch on the fp32 matrix pipe (round 3's kernels; HVN_X3=0): what the bf16x3 kernel buys
                os.environ["HVN_X3"] = "0"
                try:
                    net_nat = make_net()
                    net_nat.engine(args.batch)
                finally:
                    os.environ.pop("HVN_X3", None)
                pipe_nat = TilePipeline(net_nat, nr_types=nt, return_centroids=True)
                for _ in range(args.warmup):
                    step(p=pipe_nat)
                dtn, _ = timed(lambda: step(p=pipe_nat), args.steps)
                variants["fp32_mfma_only"] = {"value": tiles_per_step_global * args.steps / dtn, "unit": "tiles/s", "ms_per_step": 1e3 * dtn / args.steps,
                                              "what": "the timed step with every conv launch on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32; HVN_X3=0): "
                                                      "round 3's arithmetic, same checkpoint, same launch schedule"}
                result["value_fp32_mfma_only"] = variants["fp32_mfma_only"]["value"]       # top level, next to `value` / `value_host_to_host`
                del pipe_nat, net_nat
                torch.cuda.empty_cache()
            if extra is None:
                for _ in range(2):
                    step(extra_maps=structured)
                dt3, _ = timed(lambda: step(extra_maps=structured), args.steps)
                variants["plus_structured_maps"] = {"value": tiles_per_step_global * args.steps / dt3, "unit": "tiles/s", "ms_per_step": 1e3 * dt3 / args.steps,
                                                    "what": "the timed step + instance separation and table of a resident batch of structured synthetic maps "
                                                            "(2..8 nuclei per 80x80), i.e. round 3's step on this round's checkpoint"}
        if world == 1 and fitted:
            # rounds 1-3's step, for comparisons across rounds: seeded random-init checkpoint on noise tiles (0 instances in the network
            # output) + instance separation of the resident batch of structured maps
            net_r3 = net_desc.create_model(mode=args.mode, nr_types=nt, input_ch=3)
            net_r3.load_state_dict(synth_state_dict(args.mode, nt, seed=0), strict=True)
            net_r3.max_batch, net_r3.compute_dtype, net_r3.launch_schedule = args.batch, args.dtype, None
            net_r3 = net_r3.to(dev).eval()
            pipe_r3 = TilePipeline(net_r3, nr_types=nt, return_centroids=True)
            noise = [torch.from_numpy(synth_tiles(args.batch, size, seed=1)).to(dev)]
            for _ in range(args.warmup):
                step(noise, structured, pipe_r3)
            dt4, _ = timed(lambda: step(noise, structured, pipe_r3), args.steps)
            variants["round3_step_random_checkpoint"] = {"value": tiles_per_step_global * args.steps / dt4, "unit": "tiles/s", "ms_per_step": 1e3 * dt4 / args.steps,
                                                         "what": "the headline step of rounds 1-3 on this round's kernels: random-init checkpoint, noise tiles, "
                                                                 "instance separation of a resident batch of structured synthetic maps (r03: 535 tiles/s)"}
            del pipe_r3, net_r3
            torch.cuda.empty_cache()
        reps, t_end = 0, time.perf_counter() + args.sustain_seconds
        fence()
        t0 = time.perf_counter()
        while True:
            for _ in range(args.steps):
                step()
            reps += args.steps
            flag = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0], device=cdev)
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() == 0.0:
                break
        fence()
        dt_s = time.perf_counter() - t0
        variants["sustained"] = {"value": tiles_per_step_global * reps / dt_s, "unit": "tiles/s", "steps": reps, "seconds": dt_s,
                                 "what": "the timed step repeated for >= %.0f s (same mode)" % args.sustain_seconds}
        result["variants"] = variants

    # ---- roofline of the dominant kernel (rank 0) ---------------------------------------------------------------------
    def roofline_of(net_, tiles0, batch, dtype, n_prof=5):
        import ctypes

        eng = net_.engine(batch)
        assert eng.n_split == 1 and eng.n_lane_streams == 0, "the roofline leg times launches on ONE stream: launch_schedule (1, 0)"
        for o, eo in zip(eng.plan.ops, eng.ops):
            o.extra["tile_form"] = int(eo.tile_n)                     # which workgroup shape / kernel form the engine picked (roofline_account)
        timed = [o for o in eng.plan.ops if o.kind in (2, 8, 6, 7)]     # CONV, CHAIN (two chained 1x1 convs), WINO_IN, WINO_OUT: the launches hvn_profile times
        torch.cuda.synchronize(dev)
        buf = (ctypes.c_double * 4096)()
        rows = []
        for _ in range(n_prof):
            L.lib().hvn_profile_enable(1)
            run_desc.infer_step_device(tiles0, net_)            # same checkpoint, plan and kernels as the timed steps, ONE launch stream
            launches = L.lib().hvn_profile_conv_ms_list(buf, 4096)
            L.lib().hvn_profile_enable(0)
            rows.append(np.array(buf[:launches]))
        per = np.median(np.stack(rows), 0)                      # per launch: median of n_prof passes
        return roofline_account(timed, per, batch, dtype, n_prof)

    def measure_traffic(n_conv):
        """HBM bytes per conv launch, measured NOW: two `rocprofv3 --pmc` child runs of this script (FETCH_SIZE and WRITE_SIZE need
        separate passes: MI355X_MICROARCH.md, PMC slots), reduced like tools/pmc_traffic.py (FETCH_SIZE x2, the guide's gfx950 correction)."""
        import shutil
        import subprocess
        import tempfile

        exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
        if exe is None:
            return None, "rocprofv3 not found"
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import pmc_traffic

        tmp = tempfile.mkdtemp(prefix="hvn_pmc_", dir="/tmp")
        dbs = {}
        tile_file = os.path.join(tmp, "tiles.json")       # the children run THIS engine's measured column-tile choices (no autotune under the counters)
        json.dump([int(o.tile_n) for o in net_rf.engine(args.batch).ops], open(tile_file, "w"))
        try:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                cmd = [exe, "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--batch", str(args.batch),
                       "--mode", args.mode, "--nr-types", str(args.nr_types), "--dtype", args.dtype]
                env = dict(os.environ, TMPDIR="/tmp", HVN_TILE_FILE=tile_file, HVN_SPLIT="1", HVN_LANES="0")
                for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                    env.pop(k, None)
                r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=args.traffic_timeout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                found = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
                if r.returncode != 0 or not found:
                    return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode(errors="replace")[-300:])
                dbs[counter] = found[0]
            red = pmc_traffic.reduce(dbs["FETCH_SIZE"], dbs["WRITE_SIZE"], n_conv)
            if red["launches"] != n_conv:
                return None, "expected %d conv dispatches under the counters, saw %d" % (n_conv, red["launches"])
            keep = os.environ.get("HVN_KEEP_PMC_TABLE")          # path: the per-launch-class traffic table of the same two passes (tools/traffic_table.py)
            if keep:
                import traffic_table
                with open(keep, "w") as fh:
                    fh.write(traffic_table.table(dbs["FETCH_SIZE"], dbs["WRITE_SIZE"], net_rf.engine(args.batch), args.batch))
            return red, None
        except subprocess.TimeoutExpired:
            return None, "PMC child run exceeded %.0f s" % args.traffic_timeout
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    if rank == 0 and not args.no_roofline:
        if net_rf is None:
            net_rf = make_net((1, 0))
        roof = roofline_of(net_rf, tiles[0], args.batch, args.dtype)
        n_conv = roof["conv_launches_per_step"]
        red, why = (None, "--no-traffic") if (args.no_traffic or world > 1) else measure_traffic(n_conv)
        if red is not None:
            roof["traffic"] = red["hbm_bytes_per_step"] / max(1, n_conv)
            roof["traffic_hbm_bytes_per_step"] = red["hbm_bytes_per_step"]
            roof["traffic_unit"] = ("HBM bytes per conv launch, mean over the step's %d launches, MEASURED IN THIS RUN: two `rocprofv3 --pmc` child "
                                    "passes of this script (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)" % n_conv)
        else:
            for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
                tpath = os.path.join(REPO, "profiles", name)
                if os.path.exists(tpath) and args.batch == 32 and args.mode == "original" and nt == 5 and args.dtype == "fp32":
                    roof["traffic"] = json.load(open(tpath))["hbm_bytes_per_step"] / max(1, n_conv)
                    roof["traffic_unit"] = "HBM bytes per conv launch QUOTED from profiles/%s (not measured in this run: %s)" % (name, why)
                    break
            else:
                roof["traffic_unit"] = "not measured (%s)" % why
        # SURVEY 8d's whole-step figure: tiles/s x the direct-convolution FLOPs of one tile (392.17 GFLOP for cfg 2) -- what the step delivers
        # in the reference's own operation count, Winograd's savings included; per GPU
        roof["algorithmic_gflop_per_tile"] = roof["algorithmic_gflop_per_step"] / args.batch
        roof["algorithmic_tflops"] = result["value"] / world * roof["algorithmic_gflop_per_tile"] / 1e3
        roof["algorithmic_tflops_what"] = ("value / n_gpus x algorithmic_gflop_per_tile: x%.2f the fp32 matrix peak (157.3; legitimate: Winograd executes %.0f of "
                                           "%.0f GFLOP per step and the products ride the bf16 pipe), %.3f of the bf16 peak"
                                           % (roof["algorithmic_tflops"] / PEAK_FP32_MATRIX_TFLOPS, roof["executed_gflop_per_step"], roof["algorithmic_gflop_per_step"],
                                              roof["algorithmic_tflops"] / PEAK_BF16_MATRIX_TFLOPS))
        result["roofline"] = roof
        result["config"]["network_share_of_step"] = result["config"]["stage_ms"]["network"] / ms_per_step if sub == 1 else None

    # ---- BASELINE cfg 3 as a driver-visible leg: fast mode, 6 types, batch 64, bf16 (rank 0, N = 1) --------------------
    if rank == 0 and world == 1 and not args.no_variants and not args.no_cfg3 and not (args.mode == "fast" and args.dtype == "bf16"):
        nt3, b3 = 6, 64
        sd3, fit3 = make_checkpoint("fast", nt3, 256)
        net3 = make_net(None, "fast", nt3, sd3, b3, "bf16")
        tiles3 = torch.from_numpy(make_tiles(b3, 256, 1)).to(dev)
        out3 = net3.engine(b3).plan.geo["out"]
        structured3 = torch.from_numpy(synth_pred_maps(b3, out3, out3, nt3, seed=100, k_lo=2, k_hi=8)[0]).to(dev)
        pipe3 = TilePipeline(net3, nr_types=nt3, return_centroids=True)
        probe3 = pipe3.submit(tiles3, to_host=True)
        pipe3.wait()
        net_inst3 = int(probe3[2].sum().item())
        extra3 = None if (fitted and net_inst3 > 0) else structured3

        def step3():
            return pipe3.submit(tiles3, extra_maps=extra3, to_host=True)

        for _ in range(2):
            step3()
        k3 = max(5, args.steps // 2)
        dt3, out3_ = timed(step3, k3)
        net3_ms, _ = med_ms(lambda: run_desc.infer_step_device(tiles3, net3), inner=2)
        roof3 = roofline_of(net3, tiles3, b3, "bf16", n_prof=3)
        result.setdefault("variants", {})["cfg3_fast_b64_bf16"] = {
            "value": b3 * k3 / dt3, "unit": "tiles/s", "steps": k3, "ms_per_step": 1e3 * dt3 / k3, "dtype": "bf16",
            "network_ms": net3_ms, "step_over_network": 1e3 * dt3 / k3 / net3_ms,
            "workload": "PanNuke 'fast' mode (256x256 -> 164x164, 6 types), batch 64 resident in HBM, bf16 activations / weights with fp32 "
                        "accumulation and fp32 logits; same step as the headline (network + epilogue + instance separation + table of %s "
                        "+ D2H), %s checkpoint" % ("the network's own output" if extra3 is None else "the network output and of a structured batch",
                                                   "fitted" if fitted else "random-init"),
            "checkpoint": fit3, "instances_in_network_output": net_inst3,
            "instances_last_step": int(out3_[2].sum().item()), "roofline": roof3}
        del pipe3, net3
        torch.cuda.empty_cache()

    # ---- BASELINE cfg 5 / cfg 4 as driver-visible legs (outside every timed region of the headline) ---------------------------------
    sys.path.insert(0, os.path.join(REPO, "tools"))
    if not args.no_variants and not args.no_train_leg and not shared_gpu:
        # training step, both phases of the two-stage schedule (/root/reference/models/hovernet/opt.py:23-142, run_desc.py:12-109):
        # every rank runs it (N > 1: the data-parallel step with the gradient slab's all-reduce), rank 0 reports
        import train_bench

        torch.cuda.empty_cache()
        legs = {}
        try:
            for ph in (0, 1):
                r_ = train_bench.measure(ph, steps=5, warmup=2, mode=args.mode, nt=nt, device=dev)
                if world == 1:          # the same step with rounds 1-5's reduce (fp32 atomics, timed weight-gradient splits): what determinism costs
                    a_ = train_bench.measure(ph, steps=3, warmup=2, mode=args.mode, nt=nt, device=dev, deterministic=False)
                    r_["atomic_reduce_ms_per_step"] = a_["ms_per_step"]
                if world > 1:
                    allr = [None] * world
                    dist.all_gather_object(allr, {k: r_[k] for k in ("ms_per_step", "forward_ms", "loss_backward_ms", "optimizer_ms", "allreduce_slab_alone_ms")})
                    r_["per_rank"] = allr
                    r_["ms_per_step"] = max(a_["ms_per_step"] for a_ in allr)
                    r_["tiles_per_s"] = world * r_["batch"] * 1000.0 / r_["ms_per_step"]
                legs["phase%d" % ph] = r_
        except Exception as e:       # an optional leg must not cost the run its headline line.  (A code error fails alike on every rank; a rank
            legs = {"error": "%s: %s" % (type(e).__name__, e)}       # failing alone leaves the others in a collective -- no guard here helps that.)
        if rank == 0:
            result.setdefault("variants", {})["train_step"] = dict(
                legs, what="BASELINE cfg 5 on %d GPU(s): one training step (forward in train mode, the reference's loss table, backward, FusedAdam) of "
                           "phase 0 (frozen encoder, batch 16 per GPU) and phase 1 (all layers, batch 4 per GPU), CoNSeP '%s' mode, %s types, "
                           "synthetic batch; 5 steps after 2 warm-up steps each; deterministic cross-workgroup sums (round 6 default; "
                           "atomic_reduce_ms_per_step = the same step with fp32 atomics and timed weight-gradient splits); first-writer stores, the decoder branches on their own "
                           "streams and floating weight gradients as each leg's first_writer_stores / branch_streams / wgrad_stream say (round 6, bit-identical to the single-stream step); N > 1: SUM all-reduce of the loss partial sums and of the "
                           "gradient slab in two buckets inside the step (ms_per_step = slowest rank)" % (world, args.mode, nt))
    if world > 1 and (args.wsi_leg or not args.no_variants) and not args.no_wsi_leg:
        # BASELINE cfg 4 on all ranks (round-5 verdict, next #4: the row-slab + halo all_to_all path of infer_wsi.py never ran in a bench):
        # per rank stage 1 (own patch rows), the halo exchange, stage 2 (owner-post-processed tiles -> rank 0's sequential merge)
        import wsi_bench

        torch.cuda.empty_cache()
        try:
            mine = wsi_bench.measure_dist(args.wsi_size, args.mode, nt, args.batch, device=dev)
        except Exception as e:       # as for the training leg: reported, not fatal (a failure on every rank alike; the gather below still pairs up)
            mine = {"error": "%s: %s" % (type(e).__name__, e)}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        if rank == 0 and any("error" in a_ for a_ in allr):
            result.setdefault("variants", {})["wsi_%dk" % (args.wsi_size // 1024)] = {"error": [a_.get("error") for a_ in allr]}
        elif rank == 0:
            s1, s2 = max(a_["stage1_s"] for a_ in allr), max(a_["stage2_s"] for a_ in allr)
            n_p = sum(a_["patches"] for a_ in allr)
            result.setdefault("variants", {})["wsi_%dk" % (args.wsi_size // 1024)] = {
                "slide": [args.wsi_size, args.wsi_size], "world_size": world, "patches": n_p, "stage1_s": s1, "patches_per_s": n_p / s1, "stage2_s": s2,
                "instances": allr[0]["instances"], "per_rank": allr,
                "what": "BASELINE cfg 4 on %d ranks: a synthetic %d^2 slide; the prediction map is OWNED by row slabs -- every rank predicts the patch rows of its "
                        "slab (stage1_own_rows_s), receives the halo rows its stage-2 tiles reach into in ONE all_to_all_single (halo_exchange_s), "
                        "post-processes the tiles whose rows it owns and sends the results to rank 0, which applies the sequential three-phase merge "
                        "(stage2_s; infer/wsi.py:449-709).  stage1_s / stage2_s = slowest rank; stage 2 runs on a structured map written into each rank's "
                        "rows (the random-init network emits no nuclei)" % (world, args.wsi_size)}
    if rank == 0 and world == 1 and (args.wsi_leg or not args.no_variants) and not args.no_wsi_leg:
        import wsi_bench

        w_ = wsi_bench.measure(args.wsi_size, args.mode, nt, args.batch, "fp32", device=dev)
        result.setdefault("variants", {})["wsi_8k"] = dict(
            w_, what="BASELINE cfg 4 scaled to one GPU: a synthetic %d^2 slide -- stage 1 = every patch through the network into the HBM-resident "
                     "prediction map (infer/wsi.py:449-709), stage 2 = tile-wise instance separation on the GPU + the three-phase merge of a "
                     "structured prediction map of the same size (40 000^2 on one GPU: profiles/r0*_wsi_40k.json)" % args.wsi_size)

    # ---- CPU baseline (rank 0, N = 1): the oracle on a bounded sample of the same tiles ------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import net_torch, process_np

        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        # torch-CPU does not scale to every hardware thread of a big host: the thread count is calibrated on the REAL work (the oracle
        # network on one tile) and reported as `cores`
        xcal = tiles_host[0][:1].permute(0, 3, 1, 2).float()
        best = (1e9, 1)
        for th in sorted({t for t in (8, 16, 24, 32, 48, 64, 96, cores) if t <= cores}):
            torch.set_num_threads(th)
            t1 = time.perf_counter()
            net_torch.forward(sd, xcal, args.mode)
            t_th = time.perf_counter() - t1
            best = min(best, (t_th, th))
            if t_th > 2.5 * best[0]:
                break
        cores = best[1]
        torch.set_num_threads(cores)
        cpu_tiles = tiles_host[0]

        # the post-processing half in the reference's process layout (infer/tile.py:232-234: one `process` call per map in a
        # ProcessPoolExecutor; round-5 verdict, weak #7): workers = the host's cores, capped at the reference's default of 16
        # (run_infer.py: --nr_post_proc_workers) and at the number of maps; the pool is created before the clock starts, as the reference
        # creates it once per run
        import concurrent.futures as cf
        import multiprocessing as mp

        host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        workers = max(1, min(16, host_cores, args.batch))
        pool = cf.ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn"))
        list(pool.map(process_np.process, [structured_np[0]] * workers, [nt] * workers, [True] * workers))     # workers up, libraries loaded

        def cpu_pass(k):
            t1 = time.perf_counter()
            x = cpu_tiles[:k].permute(0, 3, 1, 2).float()
            pm = net_torch.infer_epilogue(net_torch.forward(sd, x, args.mode)).numpy()
            t2 = time.perf_counter()
            maps = list(pm) + (list(structured_np[:k]) if extra is not None else [])            # what the GPU step post-processes
            futs = [pool.submit(process_np.process, m, nt, True) for m in maps]
            for f in futs:
                f.result()
            return t2 - t1, time.perf_counter() - t2

        one = sum(cpu_pass(1))
        k = int(max(1, min(args.batch, round(args.cpu_seconds / max(one, 1e-3)))))
        net_s, pp_s = cpu_pass(k)
        pool.shutdown()
        result["cpu_baseline"] = {"value": k / (net_s + pp_s), "unit": "tiles/s", "cores": cores, "kind": "port",
                                  "postproc_workers": min(workers, k),
                                  "sample": "%d of the same %d tiles: oracle/net_torch.py (torch-CPU fp32, %d threads) %.2f s + "
                                            "oracle process() restatement (hvn_oracle.c + process_np.py) on the %d network maps%s in a "
                                            "ProcessPoolExecutor of %d workers, one map per call like infer/tile.py:232-234, %.2f s"
                                            % (k, args.batch, cores, net_s, k, " and %d structured maps" % k if extra is not None else "", workers, pp_s),
                                  "network_s_per_tile": net_s / k, "postproc_s_per_tile": pp_s / k}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()          # rank 0 ran the single-rank legs (stage split, roofline) after the last collective of the others
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

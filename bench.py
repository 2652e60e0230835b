#!/usr/bin/env python
"""bench.py -- tiles/sec of the HoVer-Net hot path on MI355X (BASELINE.json metric).

One *step* = one pass of the whole hot path over one batch of synthetic input already resident
in HBM: 32 uint8 270x270 tiles -> HIP network (original mode, 5 types, fp32) -> infer_step
epilogue -> on-GPU instance separation (Sobel/threshold/CC/watershed) + per-instance table, the
latter on a side stream so that it overlaps the next step's network (hover_net_amd/pipeline.py).
No host round trip inside the step.  N > 1: one process per GPU (torch.distributed / RCCL for
the barrier and the max-over-ranks clock only); tiles are independent units, so every rank
processes its own batches and there is no data-path collective ("weak" scaling).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fp32-MFMA
implicit-GEMM conv): algorithmic conv FLOPs of one step / summed duration of that step's conv
launches, measured with HIP events on the launch stream in an extra, untimed step.
`cpu_baseline` (N=1 only) times the CPU oracle (torch fp32 restatement + C post-proc port) on
a bounded sample of the same workload on this box's host cores.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--mode", default="original")
    ap.add_argument("--nr-types", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--quiet-net-output", action="store_true",
                    help="bias the NP head of the random-init checkpoint towards background, so that the network's own output "
                         "holds no nuclei and the instance-separation load of the step comes from the structured maps only "
                         "(the 'fast'-mode random init otherwise emits tile-filling blobs, the flood's worst case)")
    ap.add_argument("--dtype", default="fp32", choices=("fp32", "bf16"),
                    help="fp32 = the headline configuration (BASELINE cfg 2); bf16 = cfg 3 (use with --mode fast --nr-types 6 --batch 64)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from hover_net_amd import lib as L
    from hover_net_amd import net_desc, post_proc, run_desc
    from hover_net_amd.synth import synth_pred_maps, synth_state_dict, synth_tiles

    nt = args.nr_types if args.nr_types > 0 else None
    size = 270 if args.mode == "original" else 256
    sd = synth_state_dict(args.mode, nt, seed=0)
    if args.quiet_net_output:
        sd["decoder.np.u0.conv.bias"] = torch.tensor([8.0, -8.0])
    net = net_desc.create_model(mode=args.mode, nr_types=nt, input_ch=3)
    net.load_state_dict(sd, strict=True)
    net.max_batch = args.batch
    net.compute_dtype = args.dtype
    net = net.to(dev).eval()
    tiles = torch.from_numpy(synth_tiles(args.batch, size, seed=1 + rank)).to(dev)  # resident in HBM
    # A random-init network emits maps without nuclei (0 instances -> the watershed has nothing to
    # flood).  So that the step carries a realistic instance-separation load it ALSO post-processes a
    # resident batch of structured synthetic maps (painted, partly touching elliptical nuclei,
    # hover_net_amd.synth.synth_pred_maps): post-proc runs twice per step, which over-counts its cost.
    # Density: CoNSeP has 24 319 nuclei in 41 images of 1000x1000 px = 3.8 per 80x80 output tile; the
    # structured maps carry 2..8 (mean 5) per 80x80, scaled by area for other output sizes.
    out_hw = net.engine(args.batch).plan.geo["out"]
    structured = torch.from_numpy(synth_pred_maps(args.batch, out_hw, out_hw, nt, seed=100 + rank, k_lo=2, k_hi=8)[0]).to(dev)

    from hover_net_amd.pipeline import TilePipeline

    # network of step i+1 (main stream) overlaps the post-processing of step i (side stream)
    pipe = TilePipeline(net, nr_types=nt, return_centroids=True)

    def step():
        return pipe.submit(tiles, extra_maps=structured)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_inst = int(out[2].sum().item())

    result = {
        "metric": "tiles/sec (%dx%d, batch %d) end-to-end incl. watershed" % (size, size, args.batch),
        "value": world * args.batch * args.steps / dt,
        "unit": "tiles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": "CoNSeP '%s' mode seg+class (NP+HV+NC, %s types), batch %d of %dx%d uint8 tiles per GPU, "
                               "random-init checkpoint (seeded%s), network + infer_step epilogue + on-GPU watershed post-proc "
                               "(of the network output AND of a resident batch of structured synthetic maps, see bench.py)"
                               % (args.mode, nt, args.batch, size, size, ", NP head biased to background" if args.quiet_net_output else ""),
                   "global_batch": world * args.batch, "instances_last_step": n_inst, "parallelism": "tile-sharded x%d" % world},
    }

    if rank == 0:
        # split of one step (untimed extra passes, torch events on the launch stream)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        pred = run_desc.infer_step_device(tiles, net)
        ev[1].record()
        post_proc.process_batch_device(structured, nr_types=nt, return_centroids=True)
        ev[2].record()
        torch.cuda.synchronize(dev)
        result["config"]["network_ms"] = ev[0].elapsed_time(ev[1])
        result["config"]["postproc_structured_ms"] = ev[1].elapsed_time(ev[2])

    if rank == 0 and not args.no_roofline:
        eng = net.engine(args.batch)
        conv_flops = sum(o.flops() for o in eng.plan.ops if o.kind == 2) * args.batch
        # single launch stream for this pass (HIP events bracket each conv launch on that stream)
        saved = (eng.n_split, eng.n_lane_streams)
        eng.n_split, eng.n_lane_streams = 1, 0
        torch.cuda.synchronize(dev)
        L.lib().hvn_profile_enable(1)
        run_desc.infer_step_device(tiles, net)
        ms = L.lib().hvn_profile_conv_ms()
        launches = L.lib().hvn_profile_conv_launches()
        L.lib().hvn_profile_enable(0)
        eng.n_split, eng.n_lane_streams = saved
        achieved = conv_flops / (ms * 1e-3) / 1e12
        # MFMA FLOPs actually issued: the 5x5 decoder convs run as Winograd F(4x4,5x5) (4 instead of 25 multiplies per
        # output), so the algorithmic rate can exceed the matrix-pipe peak; the executed rate cannot
        exec_flops = sum(o.extra.get("exec_flops", o.flops()) for o in eng.plan.ops if o.kind == 2) * args.batch
        executed = exec_flops / (ms * 1e-3) / 1e12
        # HBM bytes of the same 140 launches from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        # (tools/pmc_traffic.py, gfx950 x2 correction on FETCH_SIZE); cannot be collected live
        traffic = None
        tpath = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and args.batch == 32 and args.mode == "original" and nt == 5 and args.dtype == "fp32":
            traffic = json.load(open(tpath))["hbm_bytes_per_step"]
        peak = PEAK_FP32_MATRIX_TFLOPS if args.dtype == "fp32" else PEAK_BF16_MATRIX_TFLOPS
        result["roofline"] = {"bound": "mfma", "kernel": "hvn_conv_igemm_f32" if args.dtype == "fp32" else "hvn_conv_igemm_bf16",
                              "achieved": achieved, "peak": peak,
                              "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                              "traffic_unit": "HBM bytes per step (all conv launches; rocprofv3 PMC, profiles/r01_pmc_traffic.json)",
                              "launches_per_step": launches, "conv_ms_per_step": ms, "conv_gflop_per_step": conv_flops / 1e9,
                              "executed": {"achieved": executed, "frac": executed / peak, "gflop_per_step": exec_flops / 1e9,
                                           "note": "MFMA FLOPs issued after Winograd F(4x4,5x5) on the 5x5 convs; `achieved` above is "
                                                   "direct-convolution (algorithmic) FLOPs over the same time, transforms included"}}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import net_torch
        from oracle import postproc as O

        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        # torch-CPU does not scale to every hardware thread of a big host: pick the thread count that
        # runs one mid-size conv fastest, and report it as `cores`
        import torch.nn.functional as F
        xx, ww = torch.randn(2, 256, 66, 66), torch.randn(256, 256, 3, 3)
        best = (1e9, 1)
        for th in sorted({t for t in (8, 16, 32, 64, 96, 128, cores) if t <= cores}):
            torch.set_num_threads(th)
            F.conv2d(xx, ww, padding=1)
            t1 = time.perf_counter()
            F.conv2d(xx, ww, padding=1)
            best = min(best, (time.perf_counter() - t1, th))
        cores = best[1]
        torch.set_num_threads(cores)
        cpu_tiles = tiles.cpu()

        def cpu_pass(k):
            x = cpu_tiles[:k].permute(0, 3, 1, 2).float()
            pm = net_torch.infer_epilogue(net_torch.forward(sd, x, args.mode)).numpy()
            return O.proc_batch(pm)

        t1 = time.perf_counter()
        cpu_pass(1)
        one = time.perf_counter() - t1
        k = int(max(1, min(args.batch, round(args.cpu_seconds / max(one, 1e-3)))))
        t1 = time.perf_counter()
        cpu_pass(k)
        cdt = time.perf_counter() - t1
        result["cpu_baseline"] = {"value": k / cdt, "unit": "tiles/s", "cores": cores, "kind": "port",
                                  "sample": "%d of the same %d tiles: oracle/net_torch.py (torch-CPU fp32, %d threads) + "
                                            "oracle/hvn_oracle.c post-proc (1 thread)" % (k, args.batch, cores)}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

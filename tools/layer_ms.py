#!/usr/bin/env python
"""Per-launch HIP-event times of the conv / Winograd launches of one cfg-2 network step (original mode, 5 types, batch 32, fp32)
through `hvn_profile_conv_ms_list` -- the per-layer table without rocprof, for kernel A/B runs on one box:
    HVN_LIB_VARIANT=pad HVN_FORCE_TILE_N=128 python tools/layer_ms.py > gpurun_out/x.txt
Prints `name kind tile_n median_us [executed TFLOP/s]` per launch (median of --reps passes) and the total; `--dtype bf16 --mode fast
--nr-types 6 --batch 64` is the cfg-3 table."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd import lib as L  # noqa: E402
from hover_net_amd import net_desc, run_desc  # noqa: E402
from hover_net_amd.plan import OP_CHAIN, OP_CONV, OP_WINO_IN, OP_WINO_OUT  # noqa: E402
from hover_net_amd.synth import synth_state_dict, synth_tiles  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--mode", default="original")
    ap.add_argument("--nr-types", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    args = ap.parse_args()
    nt = args.nr_types if args.nr_types > 0 else None
    net = net_desc.create_model(mode=args.mode, nr_types=nt, input_ch=3)
    net.load_state_dict(synth_state_dict(args.mode, nt, seed=0), strict=True)
    net.max_batch = args.batch
    net.compute_dtype = args.dtype
    net.launch_schedule = (1, 0)          # one launch stream: every launch is timed alone
    net = net.to("cuda").eval()
    win = 270 if args.mode == "original" else 256
    tiles = torch.from_numpy(synth_tiles(args.batch, win, seed=1)).to("cuda")
    for _ in range(2):
        run_desc.infer_step_device(tiles, net)
    torch.cuda.synchronize()
    eng = net.engine(args.batch)
    marked = [(o.name + (" [x3-%d]" % o.extra["x3"] if o.extra.get("x3") else ""), o.kind, eng.ops[i].tile_n if o.kind in (OP_CONV, OP_CHAIN) else 0,
               o.extra.get("exec_flops", o.flops()) * args.batch if o.kind in (OP_CONV, OP_CHAIN) else 0.0)
              for i, o in enumerate(eng.plan.ops) if o.kind in (OP_CONV, OP_CHAIN, OP_WINO_IN, OP_WINO_OUT)]
    buf = (ctypes.c_double * 4096)()
    rows = []
    for _ in range(args.reps):
        L.lib().hvn_profile_enable(1)
        run_desc.infer_step_device(tiles, net)
        n = L.lib().hvn_profile_conv_ms_list(buf, 4096)
        L.lib().hvn_profile_enable(0)
        rows.append(np.array(buf[:n]))
    ms = np.median(np.stack(rows), 0)
    names = marked if len(marked) == len(ms) else [("launch%d" % i, -1, 0, 0.0) for i in range(len(ms))]
    for (name, kind, tn, fl), t in zip(names, ms):
        print("%-62s %d %3d %9.1f" % (name, kind, tn, t * 1e3) + ("   %7.1f TFLOP/s" % (fl / t / 1e9) if fl else ""))
    print("TOTAL variant=%s force=%s slots64=%s cost64=%s launches=%d conv_ms=%.3f" % (
        os.environ.get("HVN_LIB_VARIANT", "-"), os.environ.get("HVN_FORCE_TILE_N", "-"), os.environ.get("HVN_WG_SLOTS_64", "-"),
        os.environ.get("HVN_NARROW_COST", "-"), len(ms), ms.sum()))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Whole-slide path timing on one MI355X (BASELINE cfg 4, scaled to one GPU): a synthetic S x S RGB slide behind
`ArraySlide` (OpenSlide is not on the box), all-tissue mask.
  stage 1  `WsiInference.raw_prediction`: chunk read -> patch gather on the GPU -> network -> scatter into the
           HBM-resident prediction map.  The random-init checkpoint's NP head is biased to background (see bench.py
           --quiet-net-output) because its raw output is not a nucleus map.
  stage 2  `WsiInference.stitch_instances` on a structured synthetic prediction map of the same size (painted nuclei at
           CoNSeP density): grid / boundary / cross tiles post-processed on the GPU + the sequential merge.
usage: python tools/wsi_bench.py [--size 8192] [--mode original] [--nr-types 5]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd import infer_wsi, net_desc  # noqa: E402
from hover_net_amd.synth import synth_pred_maps, synth_state_dict  # noqa: E402


def measure(size=8192, mode="original", nt=5, batch=32, dtype="fp32", skip_stage1=False, device="cuda"):
    """-> dict (see the module docstring): stage 1 = network over every patch of a synthetic size x size slide into the HBM-resident
    prediction map, stage 2 = tile-wise instance separation + the three-phase merge of a structured prediction map of the same size."""
    S = size
    sd = synth_state_dict(mode, nt, seed=0)
    sd["decoder.np.u0.conv.bias"] = torch.tensor([8.0, -8.0])
    net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3)
    net.load_state_dict(sd, strict=True)
    net.max_batch = batch
    net.compute_dtype = dtype
    net = net.to(device).eval()
    rng = np.random.default_rng(0)
    tile = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    slide = infer_wsi.TiledSlide(tile, (S, S))
    wsi = infer_wsi.WsiInference(net, nr_types=nt, batch_size=batch)
    mask = np.ones((S // 32, S // 32), np.uint8)
    wsi.raw_prediction(infer_wsi.TiledSlide(tile, (1024, 1024)), np.ones((32, 32), np.uint8))   # warm-up (plan, arena)
    torch.cuda.synchronize()
    t1 = float("nan")
    if not skip_stage1:
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        pm = wsi.raw_prediction(slide, mask)
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t0
        del pm
    _c, pinfo = infer_wsi.get_chunk_patch_info(np.array([S, S]), wsi.chunk_shape, wsi.pin, wsi.pout)
    n_patch = int(pinfo.shape[0])
    peak1 = torch.cuda.max_memory_allocated() / 2 ** 30
    # stage 2 on a structured map: 512^2 painted blocks tiled over the slide (nuclei at CoNSeP density: 3.8 per 80^2)
    # (assembled on the device: the 40 000^2 map is 25.6 GB)
    blk = torch.from_numpy(synth_pred_maps(4, 512, 512, nt, seed=3, k_lo=2, k_hi=6)[0]).to(device)
    reps = S // 512 + 1
    full = torch.empty((S, S, blk.shape[-1]), dtype=torch.float32, device=device)
    for r in range(reps):
        for c in range(reps):
            y0, x0 = r * 512, c * 512
            h, w = min(512, S - y0), min(512, S - x0)
            if h > 0 and w > 0:
                full[y0:y0 + h, x0:x0 + w] = blk[(r + c) % 4][:h, :w]
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    wsi.timing = {}
    t0 = time.perf_counter()
    inst_map, info = wsi.stitch_instances(full, mask)
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    grid, boundary, cross = infer_wsi.get_tile_info(np.array([S, S]), wsi.tile_shape, wsi.ambiguous_size)
    out = {"slide": [S, S], "mode": mode, "dtype": dtype, "patches": n_patch, "stage1_s": t1, "patches_per_s": n_patch / t1,
           "stage2_s": t2, "tiles": [int(grid.shape[0]), int(boundary.shape[0]), int(cross.shape[0])], "instances": len(info),
           "mpix_per_s_stage2": S * S / 1e6 / t2, "stage2_breakdown_s": wsi.timing, "peak_hbm_gib_stage1": peak1,
           "peak_hbm_gib_stage2": torch.cuda.max_memory_allocated() / 2 ** 30}
    del full, inst_map, info, wsi, net
    torch.cuda.empty_cache()
    return out


def _structured_rows(blk, lo, hi, W, device):
    """Rows [lo, hi) of the structured S x W prediction map `measure` assembles (512^2 painted blocks, block (r + c) % 4 at block row r,
    block column c): a function of ABSOLUTE coordinates, so every rank builds its own slab + halo rows without communication."""
    out = torch.empty((hi - lo, W, blk.shape[-1]), dtype=torch.float32, device=device)
    for r in range(lo // 512, (hi - 1) // 512 + 1):
        y0, y1 = max(lo, r * 512), min(hi, (r + 1) * 512)
        for c in range(-(-W // 512)):
            x0, x1 = c * 512, min(W, (c + 1) * 512)
            out[y0 - lo:y1 - lo, x0:x1] = blk[(r + c) % 4][y0 - r * 512:y1 - r * 512, :x1 - x0]
    return out


def measure_dist(size=8192, mode="original", nt=5, batch=32, device="cuda"):
    """BASELINE cfg 4 on N ranks (torch.distributed initialised, one rank per GPU): `WsiInference.raw_prediction` -- every rank predicts the
    patch rows of ITS row slab of the map and receives its halo rows in ONE all_to_all -- then `stitch_instances` on the slab-resident map
    (tiles post-processed by the owner of their rows, results to rank 0, which applies the sequential three-phase merge).  Stage 2 runs on
    a structured map (`measure`'s: the random-init network emits no nuclei), written into each rank's slab + halo rows locally.
    Returns this rank's dict; the caller gathers them (bench.py: variants.wsi_Nk.per_rank).  Reference: infer/wsi.py:449-709."""
    import torch.distributed as dist

    S = size
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
    sd = synth_state_dict(mode, nt, seed=0)
    sd["decoder.np.u0.conv.bias"] = torch.tensor([8.0, -8.0])
    net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3)
    net.load_state_dict(sd, strict=True)
    net.max_batch = batch
    net = net.to(device).eval()
    tile = np.random.default_rng(0).integers(0, 256, (512, 512, 3), dtype=np.uint8)
    slide = infer_wsi.TiledSlide(tile, (S, S))
    wsi = infer_wsi.WsiInference(net, nr_types=nt, batch_size=batch)
    mask = np.ones((S // 32, S // 32), np.uint8)
    wsi.raw_prediction(infer_wsi.TiledSlide(tile, (1024 * max(1, world), 1024)), np.ones((32 * max(1, world), 32), np.uint8))   # warm-up (plan, arena, the collective)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wsi.timing = {}
    t0 = time.perf_counter()
    pm = wsi.raw_prediction(slide, mask, as_slab=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    own_patches = int(wsi.stage1_patches)
    lo, hi = pm.rows
    blk = torch.from_numpy(synth_pred_maps(4, 512, 512, nt, seed=3, k_lo=2, k_hi=6)[0]).to(device)
    pm.t.copy_(_structured_rows(blk, lo, hi, S, device))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    inst_map, info = wsi.stitch_instances(pm, mask, shape=(S, S))
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    out = {"rank": rank, "stage1_s": t1, "stage1_own_rows_s": wsi.timing.get("stage1_own_rows_s"), "halo_exchange_s": wsi.timing.get("halo_exchange_s", 0.0),
           "halo_rows": wsi.timing.get("halo_rows", 0), "map_rows_resident": int(wsi.map_rows_resident), "patches": own_patches,
           "stage2_s": t2, "merge_s": wsi.timing.get("merge_s"), "instances": len(info) if info is not None else None}
    del pm, inst_map, info, wsi, net
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--mode", default="original")
    ap.add_argument("--nr-types", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--skip-stage1", action="store_true")
    args = ap.parse_args()
    print(json.dumps(measure(args.size, args.mode, args.nr_types if args.nr_types > 0 else None, args.batch, args.dtype, args.skip_stage1)))


if __name__ == "__main__":
    main()

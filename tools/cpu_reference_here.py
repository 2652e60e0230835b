#!/usr/bin/env python
"""The REFERENCE's own CPU path timed in the build container (the python reference under /root/reference cannot travel to the
GPU box, so bench.py's `cpu_baseline` there is the oracle port; this script puts the real thing beside it).

Process layout of the reference minus the GPU (infer/tile.py:232-234, 308-386): the network forward on torch-CPU in the main
process (`models/hovernet/net_desc.py:101-145`, imported unmodified, + the epilogue lines of run_desc.py:185-194), then
`models/hovernet/post_proc.process` per tile in a ProcessPoolExecutor under /opt/conda/bin/python3.9 (real scipy / scikit-image;
cv2 = oracle/cv2_shim).  Same tiles / weights as bench.py (seed 0 weights, seed 1 tiles, structured maps seed 100).
usage: python tools/cpu_reference_here.py [--tiles 16] [--out profiles/r02_cpu_reference_container.json]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.modules["cv2"] = types.ModuleType("cv2")

POOL = r'''
import sys, time, numpy as np
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, sys.argv[1] + "/oracle/cv2_shim"); sys.path.insert(0, "/root/reference")
import models.hovernet.post_proc as pp
assert pp.__file__.startswith('/root/reference/'), pp.__file__
maps = np.load(sys.argv[2]); workers = int(sys.argv[3]); nt = int(sys.argv[4])
def run(m):
    inst, info = pp.process(m, nr_types=nt, return_centroids=True)
    return len(info)
if __name__ == "__main__":
    with ProcessPoolExecutor(workers) as ex:
        list(ex.map(run, list(maps[:workers])))          # pool start-up is not timed
        t0 = time.perf_counter(); n = sum(ex.map(run, list(maps))); dt = time.perf_counter() - t0
    print(dt, n)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r02_cpu_reference_container.json"))
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.nn.functional as F

    import models.hovernet.net_desc as nd              # the reference, unmodified
    assert nd.__file__.startswith("/root/reference/"), nd.__file__   # never the in-tree `models/` shim package
    from hover_net_amd.synth import synth_pred_maps, synth_state_dict, synth_tiles

    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    net = nd.create_model(mode="original", nr_types=5, input_ch=3)
    net.load_state_dict(synth_state_dict("original", 5, seed=0), strict=True)
    net.eval()
    tiles = torch.from_numpy(synth_tiles(args.tiles, 270, seed=1))
    with torch.no_grad():
        net(tiles[:1].permute(0, 3, 1, 2).float())      # warm-up
        t0 = time.perf_counter()
        outs = []
        for i in range(0, args.tiles, 4):                # run_infer.py's CPU-feasible batch
            pred = net(tiles[i:i + 4].permute(0, 3, 1, 2).contiguous().float())
            pred = {k: v.permute(0, 2, 3, 1).contiguous() for k, v in pred.items()}          # run_desc.py:185-194
            pred["np"] = F.softmax(pred["np"], dim=-1)[..., 1:]
            tp = torch.argmax(F.softmax(pred["tp"], dim=-1), dim=-1, keepdim=True).float()
            outs.append(torch.cat([tp, pred["np"], pred["hv"]], -1).numpy())
        net_s = time.perf_counter() - t0
    structured = synth_pred_maps(args.tiles, 80, 80, 5, seed=100, k_lo=2, k_hi=8)[0]
    maps = np.concatenate(outs + [structured])           # what the GPU step post-processes: network maps + structured maps
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "maps.npy"), maps)
        open(os.path.join(d, "pool.py"), "w").write(POOL)
        r = subprocess.run(["/opt/conda/bin/python3.9", "-W", "ignore", os.path.join(d, "pool.py"), REPO, os.path.join(d, "maps.npy"), str(cores), "5"],
                           capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        pp_s, n_inst = r.stdout.split()
    out = {"kind": "reference", "where": "build container (no GPU)", "cores": cores, "tiles": args.tiles,
           "network_s": net_s, "postproc_s": float(pp_s), "instances": int(n_inst),
           "tiles_per_s": args.tiles / (net_s + float(pp_s)), "network_tiles_per_s": args.tiles / net_s,
           "what": "reference net_desc.HoVerNet.forward on torch-CPU (%d threads, batch 4) + run_desc epilogue, then reference post_proc.process "
                   "(scipy 1.7.1 / scikit-image 0.18.3, cv2 = oracle/cv2_shim) on the %d network maps and %d structured maps in a "
                   "ProcessPoolExecutor(%d)" % (cores, args.tiles, args.tiles, cores)}
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

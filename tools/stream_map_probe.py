#!/usr/bin/env python
"""Does the ORDER in which a process creates its HIP streams change the concurrency the inference engine's launch schedule gets?
HIP multiplexes streams onto a few hardware queues; two lanes of one step on the same queue run one after the other.
usage: python tools/stream_map_probe.py K   -- creates K idle streams first, then times the cfg-2 network step (batch 32) on the default
schedule and on the single-stream schedule; prints one line."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hover_net_amd import net_desc, run_desc  # noqa: E402
from hover_net_amd.synth import synth_state_dict, synth_tiles  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    idle = [torch.cuda.Stream() for _ in range(k)]
    for s in idle:                      # make sure the runtime really instantiates them
        with torch.cuda.stream(s):
            torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    out = []
    scheds = (None, (1, 0))
    if len(sys.argv) > 2:               # e.g. "2,1 1,2 1,3 2,0": sub-batches,lanes per schedule
        scheds = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[2:])
    for sched in scheds:
        mode, nt, bs = os.environ.get("PROBE_MODE", "original"), int(os.environ.get("PROBE_TYPES", "5")), int(os.environ.get("PROBE_BATCH", "32"))
        net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3)
        net.load_state_dict(synth_state_dict(mode, nt, seed=0), strict=True)
        net.max_batch = bs
        net.compute_dtype = os.environ.get("PROBE_DTYPE", "fp32")      # PROBE_MODE=fast PROBE_TYPES=6 PROBE_BATCH=64 PROBE_DTYPE=bf16: cfg 3
        if sched is not None:
            net.launch_schedule = sched
        net = net.to("cuda").eval()
        tiles = torch.from_numpy(synth_tiles(bs, 270 if mode == "original" else 256, seed=1)).to("cuda")
        for _ in range(3):
            run_desc.infer_step_device(tiles, net)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                run_desc.infer_step_device(tiles, net)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        out.append(best * 1e3)
        del net
    from hover_net_amd import engine as E
    pick = {k_[1:3]: (v[0], {o: round(m, 2) for o, m in v[1].items()}) for k_, v in E._STREAM_CHOICE.items()}
    if len(sys.argv) > 2:
        print("idle streams created first: %d   " % k + "   ".join("schedule %s %.2f ms" % (a, t) for a, t in zip(sys.argv[2:], out)) + "   pool offsets: %s" % pick, flush=True)
        return
    print("idle streams created first: %d   network step default schedule %.2f ms   single stream %.2f ms   pool offset timed at engine build: %s" % (
        k, out[0], out[1], pick), flush=True)


if __name__ == "__main__":
    main()

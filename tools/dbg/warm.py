"""Step time over time after engine build: does the timed region start in a transient?  usage: warm.py fitted|random"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hover_net_amd import net_desc, synth_fit
from hover_net_amd.pipeline import TilePipeline
from hover_net_amd.synth import synth_state_dict, synth_tiles
mode = sys.argv[1]
t00 = time.perf_counter()
if mode == "fitted":
    tnet, _ = synth_fit.fit("original", 5, steps=200, batch=8, lr=1e-3, seed=0, init="synth", density=synth_fit.consep_density(270))
    sd = {k: v.detach().cpu().clone() for k, v in tnet.state_dict().items()}
    tnet._train_engine = None; del tnet; torch.cuda.empty_cache()
    tiles = torch.from_numpy(synth_fit.painted_tiles(32, 270, 1, *synth_fit.consep_density(270))[0]).cuda()
else:
    sd = synth_state_dict("original", 5, seed=0)
    tiles = torch.from_numpy(synth_tiles(32, 270, seed=1)).cuda()
net = net_desc.create_model(mode="original", nr_types=5, input_ch=3)
net.load_state_dict(sd, strict=True)
net = net.to("cuda").eval()
pipe = TilePipeline(net, nr_types=5, return_centroids=True)
pipe.submit(tiles, to_host=True); pipe.wait()
print("setup %.1f s" % (time.perf_counter() - t00), flush=True)
out = []
t_start = time.perf_counter()
for blk in range(90):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        pipe.submit(tiles, to_host=True)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t_start, (time.perf_counter() - t0) / 5 * 1e3))
print(mode, " ".join("%.1fs:%.2f" % (t, ms) for t, ms in out[::3]))

import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hover_net_amd import net_desc, run_desc
from hover_net_amd.synth import synth_state_dict, synth_tiles
sd = synth_state_dict("fast", 6, seed=2)
tiles = torch.from_numpy(synth_tiles(64, 256, seed=3))
for x3, split, lanes in [("0", "1", "0"), ("0", "2", "2"), ("6", "1", "0"), ("6", "2", "0"), ("6", "1", "2"), ("6", "2", "2")]:
    os.environ["HVN_X3"], os.environ["HVN_SPLIT"], os.environ["HVN_LANES"] = x3, split, lanes
    net = net_desc.create_model(mode="fast", nr_types=6, input_ch=3)
    net.load_state_dict(sd, strict=True)
    net.max_batch = 64
    net = net.to("cuda").eval()
    full = run_desc.infer_step_device(tiles, net).cpu().clone()
    full2 = run_desc.infer_step_device(tiles, net).cpu().clone()
    res = []
    for i in (0, 31, 32, 63):
        one = run_desc.infer_step_device(tiles[i:i + 1], net).cpu()
        d = (one[0] - full[i]).abs()
        res.append("%d:%s(%.2e,%d px)" % (i, "eq" if torch.equal(one[0], full[i]) else "DIFF", float(d.max()), int((d.amax(-1) > 0).sum())))
    print("x3=%s split=%s lanes=%s repeat-equal=%s  %s" % (x3, split, lanes, torch.equal(full, full2), " ".join(res)), flush=True)
    del net
    torch.cuda.empty_cache()

"""Generate tests/golden/net_*.npz with the REFERENCE's own network code.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_net.py

Imports /root/reference/models/hovernet/net_desc.py unmodified (an empty `cv2` module is
put in sys.modules because net_utils.py:11 -> config.py imports it without using it),
loads the seeded synthetic checkpoint of hover_net_amd.synth.synth_state_dict with
strict=True, runs the reference forward + the infer_step epilogue lines
(run_desc.py:185-194) on torch-CPU fp32, and stores input seed + outputs.  These pin
oracle/net_torch.py (tests/test_oracle_net.py) and the HIP path (tests/test_gpu_net.py).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from refimport import out_dir, ref_import, selected, use_reference  # noqa: E402

use_reference()
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
ref_net = ref_import("models.hovernet.net_desc")  # the reference, unmodified (asserted to live under /root/reference)
import torch.nn.functional as F  # noqa: E402
from collections import OrderedDict  # noqa: E402
from hover_net_amd.synth import synth_state_dict, synth_tiles  # noqa: E402

CASES = {  # name: (mode, nr_types, weight seed, tile seed, n tiles, stored crop)
    "orig5": ("original", 5, 3, 5, 1, None),
    "origseg": ("original", None, 4, 6, 1, None),
    "fast6": ("fast", 6, 7, 8, 1, 64),
}
torch.set_num_threads(8)
out_dir = out_dir()
for name in selected(CASES):
    mode, nt, wseed, tseed, n, crop = CASES[name]
    net = ref_net.create_model(mode=mode, nr_types=nt, input_ch=3).eval()
    net.load_state_dict(synth_state_dict(mode, nt, seed=wseed), strict=True)
    size = 270 if mode == "original" else 256
    tiles = synth_tiles(n, size, seed=tseed)
    x = torch.from_numpy(tiles).type(torch.float32).permute(0, 3, 1, 2).contiguous()  # run_desc.py:176-177
    with torch.no_grad():
        pred = net(x)
        logits = {k: v.numpy().copy() for k, v in pred.items()}
        pred = OrderedDict([[k, v.permute(0, 2, 3, 1).contiguous()] for k, v in pred.items()])
        pred["np"] = F.softmax(pred["np"], dim=-1)[..., 1:]
        if "tp" in pred:
            pred["tp"] = torch.argmax(F.softmax(pred["tp"], dim=-1), dim=-1, keepdim=True).type(torch.float32)
        pmap = torch.cat(list(pred.values()), -1).numpy()
    if crop:
        o = (pmap.shape[1] - crop) // 2
        logits = {k: v[:, :, o:o + crop, o:o + crop] for k, v in logits.items()}
        pmap = pmap[:, o:o + crop, o:o + crop]
    np.savez_compressed(os.path.join(out_dir, "net_%s.npz" % name), mode=mode, nr_types=-1 if nt is None else nt, wseed=wseed,
                        tseed=tseed, n=n, crop=-1 if crop is None else crop, pred_map=pmap,
                        **{"logits_" + k: v for k, v in logits.items()})
    print(name, {k: v.shape for k, v in logits.items()}, pmap.shape)

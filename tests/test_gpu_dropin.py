"""-m gpu: the import-by-name boundary end to end on the GPU (the reference tree does not exist on the GPU box, so the
load sequence of /root/reference/infer/base.py:56-78 is restated line for line; tests/test_dropin_manager.py runs the
reference's own `InferManager.__load_model` against the same shims in the build container)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
REPO, tmp = sys.argv[1], sys.argv[2]
sys.path.insert(0, REPO)
from importlib import import_module
import numpy as np, torch
from hover_net_amd.synth import synth_pred_maps, synth_state_dict, synth_tiles
method = {"model_args": {"nr_types": 5, "mode": "original"}, "model_path": os.path.join(tmp, "ckpt.tar")}
sd = synth_state_dict("original", 5, seed=3)
torch.save({"desc": sd}, method["model_path"])
# ---- infer/base.py:61-77 ---------------------------------------------------------------------------------
model_desc = import_module("models.hovernet.net_desc")
net = getattr(model_desc, "create_model")(**method["model_args"])
net.load_state_dict(torch.load(method["model_path"])["desc"], strict=True)
net = torch.nn.DataParallel(net)
net = net.to("cuda")
run_step_fn = getattr(import_module("models.hovernet.run_desc"), "infer_step")
run_step = lambda input_batch: run_step_fn(input_batch, net)
post_proc_func = getattr(import_module("models.hovernet.post_proc"), "process")
# ---- infer/tile.py:308 + :137 ---------------------------------------------------------------------------
tiles = torch.from_numpy(synth_tiles(3, 270, seed=4))
out = run_step(tiles)
assert isinstance(out, np.ndarray) and out.dtype == np.float32 and out.shape == (3, 80, 80, 4)
from hover_net_amd import net_desc, post_proc, run_desc
direct = net_desc.create_model(mode="original", nr_types=5, input_ch=3)
direct.load_state_dict(sd, strict=True)
want = run_desc.infer_step(tiles, direct.to("cuda").eval())
assert np.array_equal(out, want)
from oracle import net_torch, process_np
ref = net_torch.infer_epilogue(net_torch.forward(sd, tiles[:1].permute(0, 3, 1, 2).float(), "original")).numpy()
assert np.abs(out[:1, ..., 1:] - ref[..., 1:]).max() <= 1e-3
pm = synth_pred_maps(1, 80, 80, 5, seed=9)[0][0]
inst, info = post_proc_func(pm, nr_types=5, return_centroids=True)
o_inst, o_info = process_np.process(pm, 5, True)
assert np.array_equal(inst, o_inst) and list(info) == list(o_info)
for k in info:
    assert info[k]["contour"].tolist() == o_info[k]["contour"].tolist() and info[k]["type"] == o_info[k]["type"]
    assert info[k]["centroid"].tolist() == o_info[k]["centroid"].tolist() and info[k]["type_prob"] == o_info[k]["type_prob"]
print("GPU_DROPIN_OK", len(info))
'''


def test_load_sequence_run_step_and_post_proc_through_the_shims(tmp_path):
    r = subprocess.run([sys.executable, "-c", SCRIPT, REPO, str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "GPU_DROPIN_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])

"""The import-by-name boundary (SURVEY 8b / VERDICT r1 item 9): the reference resolves its model plugins with
`import_module("models.hovernet.net_desc" | ".run_desc" | ".post_proc")` (/root/reference/infer/base.py:56-78).  With this
repository's root ahead of the reference's on sys.path those names are the in-tree shims `models/hovernet/*.py`.

CPU half (build container, where /root/reference exists): the reference's OWN `InferManager.__load_model` is run, unmodified,
against the shims -- create_model(**model_args), torch.load, convert_pytorch_checkpoint ('module.' keys), load_state_dict
(strict=True), nn.DataParallel wrap, run_step / post_proc_func binding.  The only patch is `DataParallel.to("cuda")` (no GPU
in the container).  Calling the bound run_step there must fail LOUDLY (no CPU fallback).
GPU half (tests/test_gpu_dropin.py, no reference needed): the same sequence restated, then run_step + post_proc_func on a
batch, compared with the direct hover_net_amd path.
Both run in a fresh interpreter: other tests import the reference's own `models` package into this process."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = r'''
import os, sys, types
REPO, REF, tmp = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, REF)
sys.path.insert(0, REPO)                       # the drop-in package wins the name `models`
for name in ("cv2", "termcolor"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["termcolor"].colored = lambda s, *a, **k: s
ia = types.ModuleType("imgaug"); ia.imgaug = types.ModuleType("imgaug.imgaug"); sys.modules["imgaug"] = ia; sys.modules["imgaug.imgaug"] = ia.imgaug
import numpy as np, torch
import models.hovernet.net_desc as nd
assert nd.__file__.startswith(REPO), nd.__file__
from hover_net_amd.synth import synth_state_dict
sd = synth_state_dict("original", 5, seed=3)
path = os.path.join(tmp, "ckpt.tar")
torch.save({"desc": {"module." + k: v for k, v in sd.items()}}, path)      # as saved from nn.DataParallel (run_train.py)
torch.nn.DataParallel.to = lambda self, *a, **k: self                         # the container has no GPU; everything else is the reference's code
from infer.base import InferManager                                            # the reference, unmodified
import infer.base
assert infer.base.__file__.startswith(REF)
mgr = InferManager(method={"model_args": {"nr_types": 5, "mode": "original"}, "model_path": path}, type_info_path=None)
net = mgr.run_step.__closure__[0].cell_contents if mgr.run_step.__closure__[0].cell_contents.__class__.__name__ == "DataParallel" else mgr.run_step.__closure__[1].cell_contents
assert isinstance(net, torch.nn.DataParallel)
import hover_net_amd.net_desc, hover_net_amd.post_proc, hover_net_amd.run_desc
assert type(net.module) is hover_net_amd.net_desc.HoVerNet and net.module.nr_types == 5 and net.module.mode == "original"
got = net.module.state_dict()
assert sorted(got.keys()) == sorted(sd.keys()), (len(got), len(sd))
bad = [k for k in sd if not torch.equal(got[k].double().flatten(), sd[k].double().flatten())]
assert not bad, bad[:5]
assert mgr.post_proc_func is hover_net_amd.post_proc.process
assert mgr.nr_types == 5 and len(mgr.type_info_dict) == 5
# ---- the writers of hover_net_amd.infer_manager / io_utils against the reference's own (infer/base.py:29-53,80-94, convert_format.py:17-49)
from hover_net_amd import infer_manager as im, io_utils
assert {k: (v[0], tuple(int(c) for c in v[1])) for k, v in mgr.type_info_dict.items()} == im.load_type_info(5, None)      # the 'hot' lookup-table colours
import json
tj = os.path.join(tmp, "type.json")
json.dump({str(k): ["t%d" % k, [10 * k, 20, 30]] for k in range(6)}, open(tj, "w"))
mgr2 = InferManager(method={"model_args": {"nr_types": 5, "mode": "original"}, "model_path": path}, type_info_path=tj)
assert mgr2.type_info_dict == im.load_type_info(5, tj)
info = {3: {"bbox": np.array([[1, 2], [30, 40]]), "centroid": np.array([20.25, 15.5]), "contour": np.array([[2, 1], [39, 1], [39, 29], [2, 29]], np.int32),
            "type_prob": 0.8125, "type": 2},
        np.int32(7): {"bbox": np.array([[5, 5], [9, 9]]), "centroid": np.array([6.5, 7.0]), "contour": np.array([[5, 5], [8, 5], [8, 8]], np.int32),
                      "type_prob": None, "type": None}}
ja, jb = os.path.join(tmp, "ref.json"), os.path.join(tmp, "mine.json")
ret_ref = mgr._InferManager__save_json(ja, info, mag=40)
ret_mine = io_utils.save_json(jb, info, mag=40)
assert open(ja).read() == open(jb).read() and ret_ref == ret_mine
import convert_format as cf                                                  # the reference's QuPath writer
cents, types = np.array([[20.25, 15.5], [6.5, 7.0]]), np.array([2, 4])
cf.to_qupath(os.path.join(tmp, "ref.tsv"), cents, types, mgr2.type_info_dict)
im.to_qupath(os.path.join(tmp, "mine.tsv"), cents, types, im.load_type_info(5, tj))
assert open(os.path.join(tmp, "ref.tsv")).read() == open(os.path.join(tmp, "mine.tsv")).read()
print("WRITERS_OK")
if not torch.cuda.is_available():
    try:
        mgr.run_step(torch.zeros(1, 270, 270, 3, dtype=torch.uint8))
    except Exception as e:
        assert "cuda" in str(e).lower() or "gfx950" in str(e) or "MI355X" in str(e), repr(e)
        print("LOUD:", type(e).__name__)
    else:
        raise SystemExit("run_step must not silently run on the CPU")
print("DROPIN_OK")
'''


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "infer", "base.py")), reason="needs the reference tree (build container only)")
def test_reference_infer_manager_loads_the_dropins(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, "-c", SCRIPT, REPO, REF, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "LOUD:" in r.stdout and "WRITERS_OK" in r.stdout


def test_shim_package_exports_the_boundary_names():
    """models.hovernet.{net_desc,run_desc,post_proc,targets,opt}: the symbols SURVEY 8b lists, bound to hover_net_amd."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import importlib\n"
            "nd = importlib.import_module('models.hovernet.net_desc'); rd = importlib.import_module('models.hovernet.run_desc')\n"
            "pp = importlib.import_module('models.hovernet.post_proc'); tg = importlib.import_module('models.hovernet.targets')\n"
            "op = importlib.import_module('models.hovernet.opt')\n"
            "import hover_net_amd.net_desc as a, hover_net_amd.run_desc as b, hover_net_amd.post_proc as c, hover_net_amd.targets as d, hover_net_amd.train as e\n"
            "assert nd.create_model is a.create_model and nd.HoVerNet is a.HoVerNet\n"
            "assert rd.infer_step is b.infer_step and rd.train_step is b.train_step and rd.valid_step is b.valid_step\n"
            "assert pp.process is c.process and tg.gen_targets is d.gen_targets and op.get_config is e.get_config\n"
            "import pickle; assert pickle.loads(pickle.dumps(pp.process)) is c.process\n"     # infer/tile.py:137 sends it to a worker by reference
            "print('OK')\n") % REPO
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]

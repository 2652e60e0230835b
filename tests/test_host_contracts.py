"""CPU: host-side contracts of the product that need no GPU -- the library must be missing LOUDLY (no CPU fallback), the
GPU entry points refuse to run off a gfx950 device, the fused optimizer recognises the training slab layout, and the
training schedule mirrors opt.py's phase list."""
import os

import numpy as np
import pytest
import torch

from hover_net_amd import lib as L


def test_missing_library_is_a_hard_error(monkeypatch):
    monkeypatch.setattr(L, "_LIB", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libhvn_hip.so")
    with pytest.raises(L.HvnError, match="no CPU fallback"):
        L.lib()


def test_gpu_entry_points_refuse_to_run_without_a_gfx950_device():
    from hover_net_amd import net_desc, post_proc, run_desc, targets
    from hover_net_amd.synth import synth_tiles
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    net = net_desc.create_model(mode="original", nr_types=None, input_ch=3).eval()
    with pytest.raises((RuntimeError, L.HvnError)):
        run_desc.infer_step(torch.from_numpy(synth_tiles(1, 270, seed=1)), net)
    with pytest.raises((RuntimeError, L.HvnError, AssertionError)):
        post_proc.process(np.zeros((80, 80, 3), np.float32))
    with pytest.raises((RuntimeError, L.HvnError, AssertionError)):
        targets.gen_targets(np.zeros((270, 270), np.int32), (80, 80))
    net.train()
    with pytest.raises((RuntimeError, L.HvnError)):
        net(torch.zeros(1, 3, 270, 270))


def test_fused_adam_recognises_the_training_slab_layout():
    from hover_net_amd.optim import FusedAdam
    w, g = torch.zeros(1000), torch.zeros(1000)
    shapes = [((8, 4, 3, 3), 0), ((16,), 320), ((4, 8, 1, 1), 384)]
    params = []
    for shape, off in shapes:
        numel = int(np.prod(shape))
        if len(shape) == 4:     # conv weights live channels-last in the slab
            co, ci, kh, kw = shape
            strides = (kh * kw * ci, 1, kw * ci, ci)
            p = torch.nn.Parameter(torch.as_strided(w, shape, strides, off))
            p.grad = torch.as_strided(g, shape, strides, off)
        else:
            p = torch.nn.Parameter(w[off:off + numel].view(shape))
            p.grad = g[off:off + numel].view(shape)
        params.append(p)
    opt = FusedAdam(params, lr=1e-4, betas=(0.9, 0.999))
    slab = opt._slab(opt.param_groups[0])
    assert slab is not None and slab[2:] == (0, 384 + 32)
    params[1].grad = torch.zeros(16)                      # a gradient outside the slab: no fused launch
    assert opt._slab(opt.param_groups[0]) is None
    params[1].grad = None                                 # parameters without a gradient do not matter
    assert opt._slab(opt.param_groups[0]) is not None


def test_training_schedule_mirrors_the_reference_phase_list():
    from hover_net_amd import train
    cfg = train.get_config(5, "original")
    p0, p1 = cfg["phase_list"]
    assert (p0["batch_size"], p1["batch_size"]) == ({"train": 16, "valid": 16}, {"train": 4, "valid": 8})     # opt.py:56,95
    assert p0["nr_epochs"] == p1["nr_epochs"] == 50
    assert p0["run_info"]["net"]["pretrained"] is None and p1["run_info"]["net"]["pretrained"] == -1
    assert p0["run_info"]["net"]["optimizer"][1] == {"lr": 1.0e-4, "betas": (0.9, 0.999)}
    assert p0["run_info"]["net"]["extra_info"]["loss"] == {"np": {"bce": 1, "dice": 1}, "hv": {"mse": 1, "msge": 1}, "tp": {"bce": 1, "dice": 1}}
    n0, n1 = p0["run_info"]["net"]["desc"](), p1["run_info"]["net"]["desc"]()
    assert (n0.freeze, n1.freeze, n0.nr_types, n0.mode) == (True, False, 5, "original")
    assert "tp" not in train.get_config(None, "fast")["phase_list"][0]["run_info"]["net"]["extra_info"]["loss"]
    batches = list(train.SyntheticLoader(2, 3, "fast", None, seed=1))
    assert len(batches) == 3 and batches[0]["img"].shape == (2, 256, 256, 3) and batches[0]["hv_map"].shape == (2, 164, 164, 2)


def test_proc_valid_step_output_matches_the_reference_formulas():
    """run_desc.py:262-333 restated patch by patch (the reference's own loop) against the vectorised product function."""
    from hover_net_amd import run_desc
    rng = np.random.default_rng(3)
    n, h, nt = 5, 16, 4
    raw = {"prob_np": [rng.random((h, h)) for _ in range(n)], "true_np": [rng.integers(0, 2, (h, h)) for _ in range(n)],
           "pred_tp": [rng.integers(0, nt, (h, h)).astype(np.float32) for _ in range(n)], "true_tp": [rng.integers(0, nt, (h, h)) for _ in range(n)],
           "pred_hv": [rng.normal(size=(h, h, 2)).astype(np.float32) for _ in range(n)], "true_hv": [rng.normal(size=(h, h, 2)).astype(np.float32) for _ in range(n)]}
    got = run_desc.proc_valid_step_output(raw, nr_types=nt)["scalar"]

    def dice_info(true, pred, label):
        true, pred = np.array(true == label, np.int32), np.array(pred == label, np.int32)
        return (pred * true).sum(), (pred + true).sum()
    inter = total = correct = 0
    for i in range(n):
        pred = np.array(raw["prob_np"][i] > 0.5, dtype=np.int32)
        a, b = dice_info(raw["true_np"][i], pred, 1)
        inter, total, correct = inter + a, total + b, correct + (pred == raw["true_np"][i]).sum()
    npx = n * h * h
    assert abs(got["np_acc"] - correct / npx) < 1e-12 and abs(got["np_dice"] - 2 * inter / (total + 1e-8)) < 1e-12
    for t in range(nt):
        inter = total = 0
        for i in range(n):
            a, b = dice_info(raw["true_tp"][i], raw["pred_tp"][i], t)
            inter, total = inter + a, total + b
        assert abs(got["tp_dice_%d" % t] - 2 * inter / (total + 1e-8)) < 1e-12
    mse = sum(((raw["pred_hv"][i] - raw["true_hv"][i]) ** 2).sum() for i in range(n)) / npx
    assert abs(got["hv_mse"] - mse) < 1e-5 * mse


def test_save_json_protocol(tmp_path):
    import json
    from hover_net_amd import io_utils
    info = {7: {"bbox": np.array([[1, 2], [5, 9]]), "centroid": np.array([4.5, 3.25]), "contour": np.array([[2, 1], [8, 1], [8, 4]], np.int32),
                "type_prob": 0.75, "type": 3}, 9: {"bbox": np.array([[0, 0], [2, 2]]), "centroid": np.array([1.0, 1.0]), "contour": None,
                                                    "type_prob": None, "type": None}}
    p = tmp_path / "x.json"
    io_utils.save_json(str(p), info, mag=40)
    d = json.load(open(p))
    assert d["mag"] == 40 and sorted(d["nuc"]) == ["7", "9"]
    assert d["nuc"]["7"] == {"bbox": [[1, 2], [5, 9]], "centroid": [4.5, 3.25], "contour": [[2, 1], [8, 1], [8, 4]], "type_prob": 0.75, "type": 3}


_REF_VALID = r'''
import sys, types, json
sys.path.insert(0, "/root/reference")
for name in ("cv2", "termcolor"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["termcolor"].colored = lambda s, *a, **k: s
import numpy as np
import models.hovernet.run_desc as ref                      # the reference, unmodified
assert ref.__file__.startswith("/root/reference")
ref.viz_step_output = lambda *a, **k: None                    # the image half needs cv2 / matplotlib colour maps: not under test
raw = dict(np.load(sys.argv[1], allow_pickle=False))
raw = {k: list(v) for k, v in raw.items()}
raw["imgs"] = [np.zeros((4, 4, 3), np.uint8)] * len(raw["true_np"])
out = ref.proc_valid_step_output(raw, nr_types=int(sys.argv[2]) if int(sys.argv[2]) > 0 else None)["scalar"]
print("SCALARS " + json.dumps({k: float(v) for k, v in out.items()}))
'''


@pytest.mark.skipif(not os.path.exists("/root/reference/models/hovernet/run_desc.py"), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("nt", [0, 4])
def test_proc_valid_step_output_equals_the_references_own_function(tmp_path, nt):
    """run_desc.py:262-333 run unmodified (viz half patched out) on the same accumulated validation outputs."""
    import json
    import subprocess
    import sys

    from hover_net_amd import run_desc

    rng = np.random.default_rng(11)
    n, h = 6, 20
    raw = {"prob_np": rng.random((n, h, h)), "true_np": rng.integers(0, 2, (n, h, h)), "pred_hv": rng.normal(size=(n, h, h, 2)).astype(np.float32),
           "true_hv": rng.normal(size=(n, h, h, 2)).astype(np.float32)}
    if nt:
        raw["pred_tp"] = rng.integers(0, nt, (n, h, h)).astype(np.float32)
        raw["true_tp"] = rng.integers(0, nt, (n, h, h))
    np.savez(tmp_path / "raw.npz", **raw)
    r = subprocess.run([sys.executable, "-c", _REF_VALID, str(tmp_path / "raw.npz"), str(nt)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg"))
    assert r.returncode == 0 and "SCALARS " in r.stdout, (r.stdout[-800:], r.stderr[-2500:])
    want = json.loads(r.stdout.split("SCALARS ", 1)[1].splitlines()[0])
    got = run_desc.proc_valid_step_output({k: list(v) for k, v in raw.items()}, nr_types=nt or None)["scalar"]
    assert sorted(got) == sorted(want)
    for k in want:
        assert abs(float(got[k]) - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (k, got[k], want[k])


_REF_OPT = r'''
import sys, types, json
sys.path.insert(0, "/root/reference")
for name in ("cv2", "termcolor", "skimage", "skimage.morphology", "tensorboardX"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["skimage"].morphology = sys.modules["skimage.morphology"]
sys.modules["termcolor"].colored = lambda s, *a, **k: s
sys.modules["tensorboardX"].SummaryWriter = object
import torch
import models.hovernet.opt as opt                              # the reference, unmodified
assert opt.__file__.startswith("/root/reference")
out = {}
for nt, mode in ((5, "original"), (None, "fast")):
    cfg = opt.get_config(nt, mode)
    phases = []
    for ph in cfg["phase_list"]:
        info = ph["run_info"]["net"]
        net = info["desc"]()
        sch = info["lr_scheduler"](torch.optim.SGD(torch.nn.Linear(1, 1).parameters(), lr=1.0))
        phases.append({"batch_size": ph["batch_size"], "nr_epochs": ph["nr_epochs"], "opt_args": {k: list(v) if isinstance(v, tuple) else v for k, v in info["optimizer"][1].items()},
                       "opt_name": info["optimizer"][0].__name__, "pretrained": info["pretrained"], "loss": info["extra_info"]["loss"],
                       "freeze": bool(net.freeze), "nr_types": net.nr_types, "mode": net.mode, "sched": [type(sch).__name__, sch.step_size, sch.gamma]})
    out["%s-%s" % (nt, mode)] = phases
print("CONFIG " + json.dumps(out))
'''


@pytest.mark.skipif(not os.path.exists("/root/reference/models/hovernet/opt.py"), reason="needs the reference tree (build container only)")
def test_get_config_equals_the_references_own_phase_list():
    """models/hovernet/opt.py:23-142 run unmodified vs hover_net_amd.train.get_config: batch sizes, epochs, Adam arguments, pretrained entries, loss table,
    freeze flags, StepLR(25, 0.1).  (The optimizer CLASS differs on purpose: FusedAdam = torch.optim.Adam's update on the parameter slab.)"""
    import json
    import subprocess
    import sys

    from hover_net_amd import train

    r = subprocess.run([sys.executable, "-c", _REF_OPT], capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg"))
    assert r.returncode == 0 and "CONFIG " in r.stdout, (r.stdout[-600:], r.stderr[-2500:])
    ref = json.loads(r.stdout.split("CONFIG ", 1)[1].splitlines()[0])
    for nt, mode in ((5, "original"), (None, "fast")):
        want = ref["%s-%s" % (nt, mode)]
        cfg = train.get_config(nt, mode, pretrained=want[0]["pretrained"])
        assert len(cfg["phase_list"]) == len(want) == 2
        for ph, w in zip(cfg["phase_list"], want):
            info = ph["run_info"]["net"]
            net = info["desc"]()
            sch = info["lr_scheduler"](torch.optim.SGD(torch.nn.Linear(1, 1).parameters(), lr=1.0))
            assert ph["batch_size"] == w["batch_size"] and ph["nr_epochs"] == w["nr_epochs"]
            assert {k: list(v) if isinstance(v, tuple) else v for k, v in info["optimizer"][1].items()} == w["opt_args"] and w["opt_name"] == "Adam"
            assert info["pretrained"] == w["pretrained"]
            branches = ("np", "hv") + (("tp",) if nt is not None else ())                 # the reference lists `tp` always and skips it without a tp branch
            assert {b: info["extra_info"]["loss"][b] for b in branches} == {b: w["loss"][b] for b in branches}
            assert (bool(net.freeze), net.nr_types, net.mode) == (w["freeze"], w["nr_types"], w["mode"])
            assert [type(sch).__name__, sch.step_size, sch.gamma] == w["sched"]

"""CPU: host-side contracts of the product that need no GPU -- the library must be missing LOUDLY (no CPU fallback), the
GPU entry points refuse to run off a gfx950 device, the fused optimizer recognises the training slab layout, and the
training schedule mirrors opt.py's phase list."""
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

"""CPU: the torch fp32 oracle (oracle/net_torch.py) against (a) golden logits produced by the
reference's own net_desc.py (oracle/make_golden_net.py) and (b) the reference itself when
/root/reference is present (this container only)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from hover_net_amd.synth import synth_state_dict, synth_tiles
from oracle import net_torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["orig5", "origseg", "fast6"]


def load_case(name):
    z = np.load(os.path.join(GOLD, "net_%s.npz" % name))
    mode = str(z["mode"])
    nt = int(z["nr_types"])
    nt = None if nt < 0 else nt
    sd = synth_state_dict(mode, nt, seed=int(z["wseed"]))
    tiles = synth_tiles(int(z["n"]), 270 if mode == "original" else 256, seed=int(z["tseed"]))
    crop = int(z["crop"])
    logits = {k[7:]: z[k] for k in z.files if k.startswith("logits_")}
    return mode, nt, sd, tiles, crop, logits, z["pred_map"]


def crop_to(a, crop, hw_axes):
    if crop < 0:
        return a
    o = (a.shape[hw_axes[0]] - crop) // 2
    sl = [slice(None)] * a.ndim
    for ax in hw_axes:
        sl[ax] = slice(o, o + crop)
    return a[tuple(sl)]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    mode, nt, sd, tiles, crop, logits, pmap = load_case(name)
    x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
    out = net_torch.forward(sd, x, mode)
    assert list(out.keys()) == (["np", "hv"] if nt is None else ["tp", "np", "hv"])
    for k, v in logits.items():
        got = crop_to(out[k].numpy(), crop, (2, 3))
        # same torch build => identical kernels; allow last-bit noise across hosts
        np.testing.assert_allclose(got, v, rtol=0, atol=2e-5)
    pm = crop_to(net_torch.infer_epilogue(out).numpy(), crop, (1, 2))
    np.testing.assert_allclose(pm[..., -3:], pmap[..., -3:], rtol=0, atol=2e-5)
    if nt is not None:
        assert (pm[..., 0] != pmap[..., 0]).mean() < 1e-3  # argmax may flip on near-ties only


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree only exists in the build container")
def test_oracle_matches_live_reference():
    # through oracle/refimport (round-5 verdict, weak #9): the in-tree `models` shim package must not be what answers to this name,
    # whatever was imported before this test ran
    from oracle import refimport
    saved_path, saved_mods = list(sys.path), {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("models", "dataloader", "misc", "run_utils")}
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.dont_write_bytecode = True
    refimport.use_reference()
    ref = refimport.ref_import("models.hovernet.net_desc")

    sd = synth_state_dict("original", None, seed=11)
    net = ref.create_model(mode="original", nr_types=None, input_ch=3).eval()
    net.load_state_dict(sd, strict=True)
    x = torch.from_numpy(synth_tiles(1, 270, seed=12)).float().permute(0, 3, 1, 2)
    with torch.no_grad():
        want = net(x)
    got = net_torch.forward(sd, x, "original")
    for k in want:
        assert torch.equal(want[k], got[k])
    # leave the interpreter as it was found: the reference's `models.*` must not shadow the in-tree shims for later tests
    for name in [n for n in sys.modules if n.split(".")[0] in ("models", "dataloader", "misc", "run_utils")]:
        del sys.modules[name]
    sys.modules.update(saved_mods)
    sys.path[:] = saved_path

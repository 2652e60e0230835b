"""oracle/train_torch.py (training-step restatement) against the goldens made by the reference's own
train-mode forward + losses + backward (oracle/make_golden_train.py)."""
import os

import numpy as np
import pytest
import torch

from hover_net_amd.synth import synth_state_dict, synth_train_batch
from oracle import train_torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_case(name):
    g = np.load(os.path.join(GOLD, "train_%s.npz" % name))
    mode, nt = str(g["mode"]), int(g["nr_types"])
    nt = None if nt < 0 else nt
    return g, mode, nt, bool(g["freeze"])


def sample_idx(numel):
    return [(numel * k) // 5 for k in (1, 2, 3, 4)]


def check_against_golden(g, res, rtol=2e-3):
    """Shared by the CPU oracle test and the GPU training test: loss terms, per-parameter gradient norms + samples,
    updated running-stat norms."""
    terms = dict(zip([str(k) for k in g["term_names"]], g["term_values"]))
    for k, v in terms.items():
        assert abs(res["terms"][k] - v) <= rtol * max(1.0, abs(v)), (k, res["terms"][k], v)
    assert abs(res["loss"] - float(g["loss"])) <= rtol * abs(float(g["loss"]))
    worst = 0.0
    for k, has, norm, smp in zip(g["grad_keys"], g["grad_has"], g["grad_norms"], g["grad_samples"]):
        gr = res["grads"].get(str(k))
        if not has:
            assert gr is None or float(gr.abs().max()) == 0.0, k
            continue
        assert gr is not None, k
        got = float(gr.double().norm())
        assert abs(got - norm) <= rtol * max(norm, 1e-6), (str(k), got, norm)
        flat = gr.reshape(-1)
        for i, s in zip(sample_idx(flat.numel()), smp):
            assert abs(float(flat[i]) - s) <= rtol * max(norm / np.sqrt(flat.numel()), abs(s)) + 1e-7, (str(k), i, float(flat[i]), s)
        worst = max(worst, abs(got - norm) / max(norm, 1e-12))
    for k, norm in zip(g["stat_keys"], g["stat_norms"]):
        got = float(res["new_stats"][str(k)].double().norm())
        assert abs(got - norm) <= 1e-4 * max(norm, 1e-6), (str(k), got, norm)
    return worst


@pytest.mark.parametrize("name", ["orig5_freeze", "orig5_full", "fastseg_full"])
def test_train_oracle_matches_reference_golden(name):
    g, mode, nt, freeze = load_case(name)
    torch.set_num_threads(8)
    sd = synth_state_dict(mode, nt, seed=int(g["wseed"]))
    batch = synth_train_batch(int(g["n"]), mode, nt, seed=int(g["bseed"]))
    res = train_torch.train_step(sd, batch, mode, nt, freeze)
    worst = check_against_golden(g, res, rtol=1e-4)
    assert worst < 1e-4

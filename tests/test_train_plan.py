"""The training lowering (hover_net_amd.train_plan: forward + backward op lists, shared / dilated gradient
buffers, freeze scoping) interpreted on the CPU in float64 must reproduce the training oracle's autograd
gradients to rounding (the oracle itself is pinned to the reference in test_oracle_train.py)."""
import pytest
import torch

import train_interp
from hover_net_amd.synth import synth_state_dict, synth_train_batch
from hover_net_amd.train_plan import TrainPlan
from oracle import train_torch


@pytest.mark.parametrize("mode,nt,freeze", [("original", 5, True), ("fast", None, False)])
def test_lowering_matches_oracle_fp64(mode, nt, freeze):
    torch.set_num_threads(8)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        sd = synth_state_dict(mode, nt, seed=3)
        batch = synth_train_batch(1, mode, nt, seed=11)
        ref = train_torch.train_step(sd, batch, mode, nt, freeze, dtype=torch.float64)
        P = TrainPlan(mode, nt, freeze)
        I = train_interp.Interp(P, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, 1)
        logits = I.forward(torch.from_numpy(batch["img"]))
        for k, v in logits.items():
            assert float((v - ref["logits"][k]).abs().max()) < 1e-9
        lg = {k: v.clone().requires_grad_(True) for k, v in logits.items()}
        total, _ = train_torch.loss_terms(lg, {k: torch.as_tensor(v) for k, v in batch.items()}, nt, torch.float64)
        total.backward()
        grads = I.backward({k: v.grad for k, v in lg.items()})
        assert set(grads) == set(k for k, v in ref["grads"].items() if v is not None) == P.trainable
        for k, g in grads.items():
            r = ref["grads"][k]
            assert float((g - r).abs().max()) <= 1e-8 * (float(r.abs().max()) + 1e-30), k
        for k, v in ref["new_stats"].items():
            assert float((I.sd[k] - v).abs().max()) < 1e-9, k
    finally:
        torch.set_default_dtype(old)


def test_freeze_scoping_counts():
    # the reference's golden run: 262 of 407 parameters receive a gradient in phase 0 (freeze), all in phase 1
    assert len(TrainPlan("original", 5, True).trainable) == 262
    assert len(TrainPlan("original", 5, False).trainable) == 407

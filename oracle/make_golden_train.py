"""Generate tests/golden/train_*.npz with the REFERENCE's own training code.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_train.py

Imports /root/reference/models/hovernet/{net_desc,utils}.py unmodified (empty `cv2` stub as in
make_golden_net.py), puts the model in train() mode, runs forward -> the loss composition of
run_desc.py:40-82 -> loss.backward() on torch-CPU fp32 with the seeded synthetic checkpoint and batch.
The only patch: utils.msge_loss builds its Sobel taps with device="cuda" (utils.py:122-133); torch.arange is
wrapped to drop that keyword so the function runs on the CPU.  Stored per case: the loss terms, the L2 norm
and 4 fixed samples of every parameter gradient, and the updated BatchNorm running statistics' norms --
enough to pin oracle/train_torch.py (tests/test_oracle_train.py) and, through it, the HIP training path.
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

import torch.nn.functional as F  # noqa: E402
from collections import OrderedDict  # noqa: E402

ref_net = ref_import("models.hovernet.net_desc")   # the reference, unmodified (asserted to live under /root/reference)
ref_utils = ref_import("models.hovernet.utils")
from hover_net_amd.synth import synth_state_dict, synth_train_batch  # noqa: E402

_arange = torch.arange
torch.arange = lambda *a, **k: _arange(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})

CASES = {  # name: (mode, nr_types, freeze, weight seed, batch seed, n)
    "orig5_freeze": ("original", 5, True, 3, 11, 2),
    "orig5_full": ("original", 5, False, 3, 12, 2),
    "fastseg_full": ("fast", None, False, 5, 13, 1),
}
LOSS = {"np": {"bce": 1, "dice": 1}, "hv": {"mse": 1, "msge": 1}, "tp": {"bce": 1, "dice": 1}}   # opt.py:47-51
FN = {"bce": ref_utils.xentropy_loss, "dice": ref_utils.dice_loss, "mse": ref_utils.mse_loss, "msge": ref_utils.msge_loss}


def sample_idx(numel):
    return [(numel * k) // 5 for k in (1, 2, 3, 4)]


if __name__ == "__main__":
    torch.set_num_threads(8)
    out_dir = out_dir()
    for name in selected(CASES):
        mode, nt, freeze, wseed, bseed, n = CASES[name]
        net = ref_net.create_model(mode=mode, nr_types=nt, input_ch=3, freeze=freeze)
        net.load_state_dict(synth_state_dict(mode, nt, seed=wseed), strict=True)
        batch = synth_train_batch(n, mode, nt, seed=bseed)
        imgs = torch.from_numpy(batch["img"]).type(torch.float32).permute(0, 3, 1, 2).contiguous()
        true_np = torch.from_numpy(batch["np_map"]).type(torch.int64)
        onehot = F.one_hot(true_np, num_classes=2).type(torch.float32)
        true = {"np": onehot, "hv": torch.from_numpy(batch["hv_map"]).type(torch.float32)}
        if nt is not None:
            true["tp"] = F.one_hot(torch.from_numpy(batch["tp_map"]).type(torch.int64), num_classes=nt).type(torch.float32)
        net.train()
        net.zero_grad()
        pred = net(imgs)
        pred = OrderedDict([[k, v.permute(0, 2, 3, 1).contiguous()] for k, v in pred.items()])
        pred["np"] = F.softmax(pred["np"], dim=-1)
        if nt is not None:
            pred["tp"] = F.softmax(pred["tp"], dim=-1)
        loss, terms = 0, {}
        for b in pred.keys():
            for lname, w in LOSS[b].items():
                args = [true[b], pred[b]] + ([onehot[..., 1]] if lname == "msge" else [])
                t = FN[lname](*args)
                terms["loss_%s_%s" % (b, lname)] = float(t)
                loss = loss + w * t
        loss.backward()
        keys, norms, samples, has = [], [], [], []
        for k, p in net.named_parameters():
            keys.append(k)
            g = p.grad
            has.append(g is not None)
            norms.append(0.0 if g is None else float(g.double().norm()))
            samples.append([0.0] * 4 if g is None else [float(g.reshape(-1)[i]) for i in sample_idx(g.numel())])
        skeys, snorms = [], []
        for k, b_ in net.named_buffers():
            if "running_" in k:
                skeys.append(k)
                snorms.append(float(b_.double().norm()))
        np.savez_compressed(os.path.join(out_dir, "train_%s.npz" % name), mode=mode, nr_types=-1 if nt is None else nt,
                            freeze=freeze, wseed=wseed, bseed=bseed, n=n, loss=float(loss),
                            term_names=np.array(list(terms.keys())), term_values=np.array(list(terms.values())),
                            grad_keys=np.array(keys), grad_has=np.array(has), grad_norms=np.array(norms),
                            grad_samples=np.array(samples), stat_keys=np.array(skeys), stat_norms=np.array(snorms))
        print(name, float(loss), terms, "params with grad:", sum(has), "/", len(has))

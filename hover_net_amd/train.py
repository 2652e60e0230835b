"""The training schedule on the HIP path: `get_config` mirrors /root/reference/models/hovernet/opt.py:23-142
(two phases: decoder-only with the encoder frozen at batch 16/GPU, then all layers at batch 4/GPU carrying the
weights over, Adam lr 1e-4 betas (0.9, 0.999), StepLR(25), 50 epochs each, the loss table of opt.py:47-51) and
`run_phases` is the data path of `TrainManager.run_once` / `RunEngine.run` (run_train.py:135-271,
run_utils/engine.py:132-204): build the net, load the previous phase's weights, step the loader through
`run_desc.train_step`, step the LR scheduler per epoch, run `valid_step` over the validation loader, write the
reference-format checkpoint `{"desc": state_dict}` per epoch.

Deliberately not rebuilt (host glue that never touches the GPU path): tensorboard / JSON logging
callbacks, visualisation, the file-list dataset and its imgaug augmentation pipeline -- any iterable of the
reference loader's batch dicts is accepted instead (`SyntheticLoader` provides seeded synthetic ones).

Multi-GPU: one process per GPU (`torchrun --nproc-per-node N`); `run_desc.train_step` all-reduces the loss sums
and the flat gradient slab over RCCL, every rank steps its own optimizer on identical gradients, so the weights
stay bit-identical across ranks without a broadcast.  Each rank reads its own shard of the loader.
"""
import os

import torch

from . import net_desc, run_desc
from .optim import FusedAdam
from .synth import synth_train_batch

LOSS_TABLE = {"np": {"bce": 1, "dice": 1}, "hv": {"mse": 1, "msge": 1}, "tp": {"bce": 1, "dice": 1}}   # opt.py:47-51


def get_config(nr_type, mode):
    """Same shape as opt.py:get_config's `phase_list` (the part the data path reads)."""
    def phase(freeze, train_bs, valid_bs, pretrained):
        return {
            "run_info": {"net": {
                "desc": lambda: net_desc.create_model(input_ch=3, nr_types=nr_type, freeze=freeze, mode=mode),
                "optimizer": [FusedAdam, {"lr": 1.0e-4, "betas": (0.9, 0.999)}],
                "lr_scheduler": lambda opt: torch.optim.lr_scheduler.StepLR(opt, 25),
                "extra_info": {"loss": {k: dict(v) for k, v in LOSS_TABLE.items() if k != "tp" or nr_type is not None}},
                "pretrained": pretrained,
            }},
            "batch_size": {"train": train_bs, "valid": valid_bs},
            "nr_epochs": 50,
        }
    return {"phase_list": [phase(True, 16, 16, None), phase(False, 4, 8, -1)],
            "run_engine": {"train": {"run_step": run_desc.train_step}, "valid": {"run_step": run_desc.valid_step}}}


class SyntheticLoader:
    """Seeded synthetic batches in the reference loader's format (dataloader/train_loader.py:109-137)."""

    def __init__(self, batch_size, steps, mode="original", nr_types=None, seed=0, rank=0, world=1):
        self.batch_size, self.steps, self.mode, self.nr_types = batch_size, steps, mode, nr_types
        self.seed, self.rank, self.world = seed, rank, world

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            b = synth_train_batch(self.batch_size, self.mode, self.nr_types, seed=self.seed + 1000 * (i * self.world + self.rank))
            yield {k: torch.from_numpy(v) for k, v in b.items()}


def _dist_info():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def run_phases(config, make_loaders, log_dir=None, nr_epochs=None, device="cuda", on_epoch=None):
    """config: get_config(...); make_loaders(phase_idx, batch_size_dict) -> {"train": iterable, "valid": iterable|None}.
    Returns the per-epoch history [{phase, epoch, lr, train: {EMA means}, valid_steps}] and the final net."""
    rank, _world = _dist_info()
    history, prev_state, net = [], None, None
    for pi, phase in enumerate(config["phase_list"]):
        info = phase["run_info"]["net"]
        net = info["desc"]()
        pre = info["pretrained"]
        if pre == -1:                       # weights of the previous phase (run_train.py:176-180)
            if prev_state is None:
                raise ValueError("phase %d asks for the previous phase's weights but there is none" % pi)
            net.load_state_dict(prev_state, strict=True)
        elif pre is not None:
            sd = torch.load(pre, map_location="cpu")
            sd = sd["desc"] if "desc" in sd else sd
            missing, unexpected = net.load_state_dict(sd, strict=False)    # ImageNet encoder: decoder keys are missing
            if unexpected:
                raise KeyError("unexpected keys in %s: %s" % (pre, unexpected[:4]))
        net = net.to(device)
        opt_cls, opt_args = info["optimizer"]
        optimizer = opt_cls(net.parameters(), **opt_args)
        scheduler = info["lr_scheduler"](optimizer)
        run_info = [{"net": {"desc": net, "optimizer": optimizer, "lr_scheduler": scheduler, "extra_info": info["extra_info"]}},
                    {"epoch": 0, "step": 0}]
        loaders = make_loaders(pi, phase["batch_size"])
        for epoch in range(nr_epochs if nr_epochs is not None else phase["nr_epochs"]):
            ema, steps = {}, 0
            for batch in loaders["train"]:
                out = run_desc.train_step(batch, run_info)
                for k, v in out["EMA"].items():                     # ScalarMovingAverage(alpha=0.95), run_utils/callbacks/base.py
                    ema[k] = v if k not in ema else 0.95 * ema[k] + 0.05 * v
                steps += 1
                run_info[1]["step"] += 1
            lr = optimizer.param_groups[0]["lr"]
            nvalid = 0
            if loaders.get("valid") is not None:
                for batch in loaders["valid"]:
                    run_desc.valid_step(batch, run_info)
                    nvalid += 1
            scheduler.step()                                         # ScheduleLr on EPOCH_COMPLETED
            run_info[1]["epoch"] += 1
            rec = {"phase": pi, "epoch": epoch, "lr": lr, "steps": steps, "train": dict(ema), "valid_steps": nvalid}
            history.append(rec)
            if log_dir is not None and rank == 0:                    # PeriodicSaver: {"desc": state_dict}
                os.makedirs(os.path.join(log_dir, "%02d" % pi), exist_ok=True)
                sd = {k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()}
                torch.save({"desc": sd, "epoch": epoch}, os.path.join(log_dir, "%02d" % pi, "net_epoch=%d.tar" % (epoch + 1)))
            if on_epoch is not None:
                on_epoch(rec)
        prev_state = {k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()}
    return history, net

"""The training schedule on the HIP path: `get_config` mirrors /root/reference/models/hovernet/opt.py:23-142
(two phases: decoder-only with the encoder frozen at batch 16/GPU, then all layers at batch 4/GPU carrying the
weights over, Adam lr 1e-4 betas (0.9, 0.999), StepLR(25), 50 epochs each, the loss table of opt.py:47-51) and
`run_phases` is `TrainManager.run_once` (run_train.py:135-271) on `hover_net_amd.run_engine` (the RunEngine / Events /
callback protocol of run_utils/engine.py:132-204): build the net, load the previous phase's weights, wire the train and valid
engines the way opt.py:96-140 does (ScalarMovingAverage, TrackLr, PeriodicSaver, TriggerEngine("valid"), ScheduleLr;
AccumulateRawOutput, ProcessAccumulatedRawOutput) and run them; checkpoints are `{"desc", "optimizer", "lr_scheduler"}`
state_dicts per epoch like the reference's PeriodicSaver.

Deliberately not rebuilt (host glue that never touches the GPU path): tensorboard / JSON logging
callbacks, visualisation, the file-list dataset and its imgaug augmentation pipeline -- any iterable of the
reference loader's batch dicts is accepted instead (`SyntheticLoader` provides seeded synthetic ones).

Multi-GPU: one process per GPU (`torchrun --nproc-per-node N`, device = LOCAL_RANK); at the start of every phase rank 0's
parameters and buffers are BROADCAST to all ranks (each process draws its own random init, and the reference runs one
process), `run_desc.train_step` then all-reduces the loss sums and the flat gradient slab over RCCL and every rank steps
its own optimizer on identical gradients, so the weights stay bit-identical across ranks.  Each rank reads its own shard
of the loader; every rank must see the same number of equally sized batches per epoch (checked; ragged last batches are
dropped when world > 1, the all-reduce would otherwise dead-lock or mis-scale the loss normalisation).
"""
import os

import torch

from . import net_desc, run_desc
from . import run_engine as RE
from .optim import FusedAdam
from .synth import synth_train_batch

LOSS_TABLE = {"np": {"bce": 1, "dice": 1}, "hv": {"mse": 1, "msge": 1}, "tp": {"bce": 1, "dice": 1}}   # opt.py:47-51


def get_config(nr_type, mode, pretrained=None):
    """Same shape as opt.py:get_config's `phase_list` (the part the data path reads).  `pretrained`: path of the encoder
    checkpoint phase 0 starts from (opt.py:53 hard-codes "../pretrained/ImageNet-ResNet50-Preact_pytorch.tar"; there is no
    network here to fetch it, so it is an argument).  Phase 0 freezes the encoder: without a pretrained encoder the decoder
    would train on a frozen RANDOM encoder, which `run_phases` refuses unless `allow_random_frozen_encoder=True`."""
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
    return {"phase_list": [phase(True, 16, 16, pretrained), phase(False, 4, 8, -1)],
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


class _DropRagged:
    """drop_last for a multi-rank run: a batch of another size would mis-scale the all-reduced loss normalisation."""

    def __init__(self, loader, batch_size):
        self.loader, self.batch_size = loader, batch_size

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        return (b for b in self.loader if int(b["img"].shape[0]) == self.batch_size)


def device_loaders(train_patches, valid_patches, mode, with_type, seed=0, device=None):
    """`make_loaders` for `run_phases` over HBM-resident patch sets (`augment.DevicePatchLoader`): the FileLoader + DataLoader pair of
    run_train.py:106-133 with the shapes of config.py (original: 270 -> 80, fast: 256 -> 164).  `*_patches`: lists of `.npy` paths or
    [P,H,W,5] arrays.  The sets are uploaded once and shared by both phases; rank / world come from torch.distributed."""
    from . import augment

    act, out = ((270, 270), (80, 80)) if mode == "original" else ((256, 256), (164, 164))
    rank, world = _dist_info()
    if device is None:
        device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    cache = {}

    def make(phase_idx, batch_size):
        out_d = {}
        for split, patches in (("train", train_patches), ("valid", valid_patches)):
            if patches is None:
                out_d[split] = None
                continue
            if split not in cache:
                cache[split] = augment.DevicePatchLoader(patches, act, out, batch_size[split], mode=split, with_type=with_type, seed=seed,
                                                         device=device, rank=rank, world=world)
            ld = cache[split]
            ld.batch_size = int(batch_size[split])      # phase 1 runs smaller batches over the same resident set
            out_d[split] = ld
        return out_d

    return make


def _dist_info():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def convert_checkpoint_keys(sd):
    """run_utils/utils.py:convert_pytorch_checkpoint: a checkpoint saved from nn.DataParallel carries 'module.' on every
    key; strip it (only when ALL keys have it, like the reference)."""
    keys = list(sd.keys())
    if keys and all(k.startswith("module.") for k in keys):
        return {k[len("module."):]: v for k, v in sd.items()}
    return sd


def broadcast_module_state(net, src=0):
    """Make every rank start a phase from rank `src`'s parameters and buffers (one flat broadcast per dtype)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors = [p.data for p in net.parameters()] + [b.data for b in net.buffers()]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for group in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src=src)
        off = 0
        for t in group:
            t.copy_(flat[off:off + t.numel()].reshape(t.shape))
            off += t.numel()


def _same_on_all_ranks(value, what, device=None):
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dev = (device if device is not None else "cuda") if dist.get_backend() == "nccl" else "cpu"      # THIS rank's GPU, never "whatever is current"
    t = torch.tensor([float(value), -float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if t[0].item() != -t[1].item():
        raise ValueError("%s differs between ranks (max %g, min %g): every rank must run the same number of equally sized steps"
                         % (what, t[0].item(), -t[1].item()))


def run_phases(config, make_loaders, log_dir=None, nr_epochs=None, device=None, on_epoch=None, allow_random_frozen_encoder=False,
               handlers=None):
    """config: get_config(...); make_loaders(phase_idx, batch_size_dict) -> {"train": iterable, "valid": iterable|None}.
    Returns the per-epoch history [{phase, epoch, lr, train: {EMA means}, valid_steps, valid: {scalars}}] and the final net.
    `handlers`: extra (event, handler) pairs for the train engine -- any object with the reference's `.run(state, event)`
    protocol (run_utils/callbacks/*), e.g. its logging callbacks."""
    rank, world = _dist_info()
    if device is None:                      # one process per GPU: the launcher's LOCAL_RANK names it
        device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.is_available() else "cuda"
    if torch.cuda.is_available() and torch.device(device).type == "cuda":
        # one process per GPU: collectives that allocate on "the current device" and the ctypes HIP launches (issued on the current
        # device's streams) must all see THIS rank's GPU
        torch.cuda.set_device(torch.device(device))
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
            sd = convert_checkpoint_keys(sd["desc"] if "desc" in sd else sd)
            missing, unexpected = net.load_state_dict(sd, strict=False)    # ImageNet encoder: decoder keys are missing
            if unexpected:
                raise KeyError("unexpected keys in %s: %s" % (pre, unexpected[:4]))
        elif getattr(net, "freeze", False) and not allow_random_frozen_encoder:
            raise ValueError("phase %d freezes the encoder but has no pretrained checkpoint (get_config(..., pretrained=PATH)): "
                             "it would train the decoder on a frozen random encoder; pass allow_random_frozen_encoder=True "
                             "to do that on purpose (synthetic benchmarks)" % pi)
        net = net.to(device)
        broadcast_module_state(net)         # before the TrainEngine re-points the parameters at its slab
        opt_cls, opt_args = info["optimizer"]
        optimizer = opt_cls(net.parameters(), **opt_args)
        scheduler = info["lr_scheduler"](optimizer)
        run_info = {"net": {"desc": net, "optimizer": optimizer, "lr_scheduler": scheduler, "extra_info": info["extra_info"]}}
        loaders = make_loaders(pi, phase["batch_size"])
        if hasattr(loaders["train"], "__len__"):
            _same_on_all_ranks(len(loaders["train"]), "phase %d: number of training batches per epoch" % pi, device)
        train_bs = int(phase["batch_size"]["train"])
        train_loader = _DropRagged(loaders["train"], train_bs) if world > 1 else loaders["train"]
        # ---- the wiring of opt.py:96-140 on run_engine.RunEngine --------------------------------------------------------------
        step_fns = config.get("run_engine", {})
        train_eng = RE.RunEngine("train", train_loader, step_fns.get("train", {}).get("run_step", run_desc.train_step), run_info)
        valid_eng = None
        train_eng.add_event_handler(RE.Events.STEP_COMPLETED, RE.ScalarMovingAverage())
        train_eng.add_event_handler(RE.Events.EPOCH_COMPLETED, RE.TrackLr())
        if log_dir is not None and rank == 0:
            os.makedirs(os.path.join(log_dir, "%02d" % pi), exist_ok=True)
            train_eng.state.logging, train_eng.state.log_dir = True, os.path.join(log_dir, "%02d" % pi)
            train_eng.add_event_handler(RE.Events.EPOCH_COMPLETED, RE.PeriodicSaver())
        if loaders.get("valid") is not None:
            valid_eng = RE.RunEngine("valid", loaders["valid"], step_fns.get("valid", {}).get("run_step", run_desc.valid_step), run_info)
            valid_eng.add_event_handler(RE.Events.STEP_COMPLETED, RE.AccumulateRawOutput())
            valid_eng.add_event_handler(RE.Events.EPOCH_COMPLETED, RE.ProcessAccumulatedRawOutput(
                lambda raw: run_desc.proc_valid_step_output(raw, nr_types=net.nr_types)))
            trig = RE.TriggerEngine("valid")
            trig.triggered_engine = valid_eng
            train_eng.add_event_handler(RE.Events.EPOCH_COMPLETED, trig)
        for extra in (handlers or ()):                      # e.g. the reference's own logging / visualisation callbacks
            train_eng.add_event_handler(*extra)

        class _Record(RE.BaseCallbacks):                    # history row per epoch, before ScheduleLr changes the rate
            def run(_self, state, event):
                nvalid = 0 if valid_eng is None else valid_eng.state.curr_epoch_step
                if valid_eng is not None:
                    valid_eng.state.curr_epoch_step = 0
                rec = {"phase": pi, "epoch": state.curr_epoch - 1, "lr": optimizer.param_groups[0]["lr"], "steps": state.curr_epoch_step,
                       "train": {k: v for k, v in state.tracked_step_output["scalar"].items() if not k.startswith("lr-")},
                       "valid_steps": nvalid, "valid": None if valid_eng is None else valid_eng.state.tracked_step_output.get("scalar")}
                state.curr_epoch_step = 0
                history.append(rec)
                if on_epoch is not None:
                    on_epoch(rec)

        train_eng.add_event_handler(RE.Events.EPOCH_COMPLETED, _Record())
        train_eng.add_event_handler(RE.Events.EPOCH_COMPLETED, RE.ScheduleLr())
        train_eng.run(nr_epochs if nr_epochs is not None else phase["nr_epochs"])
        prev_state = {k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()}
    return history, net

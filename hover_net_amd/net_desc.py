"""Drop-in for `models.hovernet.net_desc` (/root/reference/models/hovernet/net_desc.py).

Same public surface -- `create_model(mode, **kwargs)`, `HoVerNet(input_ch, nr_types,
freeze, mode)` with attributes `.mode .freeze .nr_types .output_ch`, the exact
`state_dict()` key set / shapes (checked with strict=True by infer/base.py:65-68), and
`forward(imgs float32 NCHW 0..255) -> OrderedDict(tp?, np, hv)` of raw logits
(net_desc.py:101-145) -- but the module holds no layer objects: parameters hang off a
key-shaped tree, and `forward` lowers them once to the fused HIP launch plan
(`hover_net_amd.plan`) executed by libhvn_hip.so.  There is no torch fallback: without
the library or off a gfx950 device `forward` raises.

In train() mode `forward` runs the training engine's forward (batch-statistics BatchNorm, running stats updated,
activations kept for the backward pass).  The returned logits are nodes of torch's autograd graph (`_TrainForward`): a loss
computed on them in torch and `loss.backward()` hand the logit gradients to the HIP backward plan, which fills the
parameters' gradients -- so the module is "autograd-capable in train mode"  although no torch op computes a
gradient.  `run_desc.train_step` does not take this detour: its losses and logit gradients are fused kernels.
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import arch


class _Node(nn.Module):
    """One path component of a checkpoint key ('d0', 'units', '0', 'conv1/bn', ...)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; HoVerNet.forward runs the fused HIP plan")


def _attach(root, key, kind, shape, gen):
    parts = key.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    leaf = parts[-1]
    if kind in ("conv", "bias", "bn_w", "bn_b"):
        t = torch.empty(shape, dtype=torch.float32)
        if kind == "conv":      # Net.weights_init (net_utils.py:18-32): Kaiming normal, fan_out, relu
            nn.init.kaiming_normal_(t, mode="fan_out", nonlinearity="relu", generator=gen)
        elif kind == "bias":    # nn.Conv2d default bias init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)), fan_in = 64
            t.uniform_(-0.125, 0.125, generator=gen)
        elif kind == "bn_w":
            t.fill_(1.0)
        else:
            t.zero_()
        node.register_parameter(leaf, nn.Parameter(t))
    elif kind == "bn_rm":
        node.register_buffer(leaf, torch.zeros(shape))
    elif kind == "bn_rv":
        node.register_buffer(leaf, torch.ones(shape))
    elif kind == "bn_nbt":
        node.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
    elif kind == "ones":
        node.register_buffer(leaf, torch.ones(shape))
    else:  # pragma: no cover
        raise KeyError(kind)


class _TrainForward(torch.autograd.Function):
    """Autograd node around the training engine: forward = HIP train-mode forward, backward = HIP backward plan.
    Inputs: the module, the images, then every trainable parameter (they make the outputs require grad and receive the
    gradients).  The engine writes the gradients into its slab, which is also the memory behind `p.grad`; autograd ADDS what
    backward() returns to `p.grad`, so the slab's previous content is put back and the new gradients are handed over as a
    copy -- `p.grad` ends up holding the running sum over the backward calls since the last zero_grad, torch's semantics."""

    @staticmethod
    def forward(ctx, net, imgs, *params):
        from . import train_engine

        teng = train_engine.engine_for(net, imgs.shape[0])
        teng.img.copy_(imgs.detach().permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8))   # the loader's uint8 pixels
        logits = teng.forward()
        ctx.teng, ctx.keys = teng, list(logits.keys())
        ctx.param_keys = [k for k, p in net.named_parameters() if p.requires_grad and k in teng.plan.trainable]
        return tuple(v.clone() for v in logits.values())

    @staticmethod
    def backward(ctx, *grads):
        teng = ctx.teng
        held = teng.gslab.clone()                 # what p.grad holds so far (the slab IS the parameters' .grad memory)
        teng.backward_from({k: g for k, g in zip(ctx.keys, grads) if g is not None})
        snap = teng.gslab.clone()
        teng.gslab.copy_(held)                    # autograd adds `snap` to it: p.grad = previous + this pass, torch's semantics
        out = tuple(teng._param_view(snap, k, teng._poff[k]) for k in ctx.param_keys)
        return (None, None) + out


BF16_CHAIN_DEFAULT = "d0d1"      # which blocks' seams the bf16 plan chains by default (set from the measurement: DESIGN section 4.8)


class HoVerNet(nn.Module):
    """Initialise HoVer-Net (interface of net_desc.py:14-99)."""

    def __init__(self, input_ch=3, nr_types=None, freeze=False, mode="original"):
        super().__init__()
        self.mode = mode
        self.freeze = freeze
        self.nr_types = nr_types
        self.output_ch = 3 if nr_types is None else 4
        assert mode == "original" or mode == "fast", \
            "Unknown mode `%s` for HoVerNet. Only support `original` or `fast`." % mode
        if input_ch != 3:
            raise ValueError("the HIP conv0 kernel is built for 3-channel (RGB) input, got input_ch=%d" % input_ch)
        for key, (kind, shape) in arch.param_table(mode, nr_types, input_ch).items():
            _attach(self, key, kind, shape, None)
        self._engine = None
        self._engine_key = None
        self.max_batch = 32
        # "fp32" (parity configuration, logits within 1e-3) or "bf16" (BASELINE cfg 3: bf16 weights / activations, fp32
        # accumulation; no reference bf16 exists, the declared tolerance is on the fp32 logits, tests/test_gpu_bf16.py)
        self.compute_dtype = os.environ.get("HVN_DTYPE", "fp32")
        # launch schedule of the inference engine: None = the engine's default (fp32: two encoder sub-batches + decoder branch
        # streams), or (n_split, n_lanes); (1, 0) = one launch stream (every launch can then be timed alone: bench.py's roofline leg)
        self.launch_schedule = None
        # how the fp32 network is lowered: "default" (bf16x3 products with six partial products, F(6x6) Winograd tiles where the
        # trained-like parity test keeps a 3 - 4x margin) or "conservative" (nine partial products, F(4x4) tiles everywhere) for
        # checkpoints with hotter activations than the qualified range (plan.build_plan); set before the first forward
        self.lowering = "default"

    def _apply(self, fn, *a, **k):
        # .to() / .cuda() / .float() replace the parameter storage: the bound plans (and the training engine's slabs
        # the parameters were re-pointed at) are rebuilt on next use
        self._engine, self._engine_key = None, None
        if getattr(self, "_train_engine", None) is not None:
            self._train_engine = None
        return super()._apply(fn, *a, **k)

    # -- plan lifetime ---------------------------------------------------------------------
    def _weights_version(self):
        # the HIP training kernels update weights / running stats behind torch's version counters: the training
        # engine bumps _train_version on every step
        return (getattr(self, "_train_version", 0),) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def engine(self, batch):
        """(Re)lower the checkpoint when the weights changed or a larger batch arrives."""
        from . import engine as E
        from . import plan as PL

        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("HoVerNet runs on MI355X only: call .to('cuda') first (no CPU fallback)")
        import os
        key = (self._weights_version(), str(dev), self.compute_dtype, self.launch_schedule, self.lowering, os.environ.get("HVN_BF16_CHAIN", BF16_CHAIN_DEFAULT))
        if self._engine is None or self._engine_key != key or batch > self._engine.max_batch:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            bf16 = self.compute_dtype == "bf16"
            # bf16: direct convolutions, no bf16x3; HVN_BF16_CHAIN = d0 | d0d1 chains those blocks' conv3 -> conv1 seams (csrc/hvn_conv_chain_bf16.hip, same bits), 0 none
            import os
            bchain = os.environ.get("HVN_BF16_CHAIN", BF16_CHAIN_DEFAULT)
            plan = PL.build_plan(sd, self.mode, self.nr_types, winograd=0 if bf16 else None,
                                 chain=(("bf16:" + bchain) if bchain not in ("0", "") else False) if bf16 else None, x3=0 if bf16 else None,
                                 lowering=self.lowering)
            self._engine = None  # free the old arena first
            ns, nl = self.launch_schedule if self.launch_schedule is not None else (None, None)
            self._engine = E.Engine(plan, max(self.max_batch, batch), dev, dtype=self.compute_dtype, n_split=ns, n_lanes=nl)
            self._engine_key = key
        return self._engine

    def forward(self, imgs):
        if self.training:
            from . import train_engine

            teng = train_engine.engine_for(self, imgs.shape[0])      # re-points the parameters at its slab on first use
            params = [p for k, p in self.named_parameters() if p.requires_grad and k in teng.plan.trainable]
            outs = _TrainForward.apply(self, imgs, *params)
            return OrderedDict(zip(teng.logits.keys(), outs))
        eng = self.engine(imgs.shape[0])
        logits, _ = eng.run(imgs)
        # fresh tensors: the engine's buffers are overwritten by the next call
        return OrderedDict((k, v.clone()) for k, v in logits.items())


def create_model(mode=None, **kwargs):
    if mode not in ["original", "fast"]:
        assert "Unknown Model Mode %s" % mode
    return HoVerNet(mode=mode, **kwargs)

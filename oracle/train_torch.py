"""Plain-PyTorch fp32 restatement of the reference TRAINING step -- TEST INFRASTRUCTURE ONLY.

One functional forward in training mode (batch-statistics BatchNorm with running-stat update, the `freeze`
grad-scoping of the reference) + the four loss terms + autograd, over a reference-format `state_dict`, so that
it travels to the GPU box where /root/reference does not exist.  Pinned against the real reference
(models.hovernet.net_desc.HoVerNet in train() mode + models.hovernet.utils losses + loss.backward()) by
tests/test_oracle_train.py in the build container and by tests/golden/train_*.npz
(oracle/make_golden_train.py).

What each block follows (paths under /root/reference/models/hovernet/):
  train_step               run_desc.py:12-109   (one-hot targets, softmax, loss sum, backward)
  forward, freeze scoping  net_desc.py:101-145 (conv0 and conv_bot always carry grad; d1..d3 are under
                           set_grad_enabled(not freeze)), net_utils.py:250-266 (inside d0 only the units are
                           scoped: its shortcut conv and blk_bna carry grad even when frozen)
  BatchNorm2d (train)      torch semantics: biased batch variance for the normalisation, running stats updated
                           with momentum 0.1 and the unbiased variance, also under no_grad
  xentropy / dice / mse / msge   utils.py:54-172
  loss weights             opt.py:47-51 (all 1)
"""
import torch
import torch.nn.functional as F

from .net_torch import EPS, RES_STRIDE, RES_UNITS, _crop, _tf_same_pad, _upsample2x

MOMENTUM = 0.1


def _bn(sd, new_stats, key, x):
    rm, rv = sd[key + ".running_mean"].clone(), sd[key + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[key + ".weight"], sd[key + ".bias"], training=True, momentum=MOMENTUM, eps=EPS)
    new_stats[key + ".running_mean"], new_stats[key + ".running_var"] = rm, rv
    return F.relu(y)


def _res_block(sd, ns, name, x, freeze):
    stride = RES_STRIDE[name]
    shortcut = F.conv2d(x, sd[name + ".shortcut.weight"], stride=stride)
    prev = x
    for i in range(RES_UNITS[name]):
        p = "%s.units.%d." % (name, i)
        with torch.set_grad_enabled(torch.is_grad_enabled() and not freeze):   # net_utils.py:258-260
            f = prev
            if i != 0:
                f = _bn(sd, ns, p + "preact/bn", f)
            f = _bn(sd, ns, p + "conv1/bn", F.conv2d(f, sd[p + "conv1.weight"]))
            s = stride if i == 0 else 1
            f = F.conv2d(_tf_same_pad(f, 3, s), sd[p + "conv2.weight"], stride=s)
            f = F.conv2d(_bn(sd, ns, p + "conv2/bn", f), sd[p + "conv3.weight"])
        prev = f + shortcut
        shortcut = prev
    return _bn(sd, ns, name + ".blk_bna.bn", prev)


def _dense_block(sd, ns, name, x, n_units):
    for i in range(n_units):
        p = "%s.units.%d." % (name, i)
        f = _bn(sd, ns, p + "preact_bna/bn", x)
        f = _bn(sd, ns, p + "conv1/bn", F.conv2d(f, sd[p + "conv1.weight"]))
        f = F.conv2d(f, sd[p + "conv2.weight"], groups=4)
        x = torch.cat([_crop(x, x.shape[2] - f.shape[2], x.shape[3] - f.shape[3]), f], dim=1)
    return _bn(sd, ns, name + ".blk_bna.bn", x)


def forward_train(sd, imgs, mode="original", freeze=False):
    """sd: state_dict whose trainable tensors have requires_grad; imgs float32 NCHW 0..255.
    -> (OrderedDict-like dict of logits tp?, np, hv; dict of updated running stats)."""
    ns = {}
    x = imgs / 255.0
    if mode == "fast":
        x = _tf_same_pad(x, 7, 1)
    x = _bn(sd, ns, "conv0.bn", F.conv2d(x, sd["conv0./.weight"]))
    d = [_res_block(sd, ns, "d0", x, freeze)]
    with torch.set_grad_enabled(not freeze):                                    # net_desc.py:108-111
        for name in ("d1", "d2", "d3"):
            d.append(_res_block(sd, ns, name, d[-1], False))
    d[3] = F.conv2d(d[3], sd["conv_bot.weight"])
    c0, c1 = (184, 72) if mode == "original" else (92, 36)
    d[0], d[1] = _crop(d[0], c0, c0), _crop(d[1], c1, c1)
    out = {}
    for b in [b for b in ("tp", "np", "hv") if ("decoder.%s.u0.conv.weight" % b) in sd]:
        p = "decoder.%s." % b
        u3 = F.conv2d(_upsample2x(d[3]) + d[2], sd[p + "u3.conva.weight"])
        u3 = F.conv2d(_dense_block(sd, ns, p + "u3.dense", u3, 8), sd[p + "u3.convf.weight"])
        u2 = F.conv2d(_upsample2x(u3) + d[1], sd[p + "u2.conva.weight"])
        u2 = F.conv2d(_dense_block(sd, ns, p + "u2.dense", u2, 4), sd[p + "u2.convf.weight"])
        u1 = _upsample2x(u2) + d[0]
        k = sd[p + "u1.conva.weight"].shape[2]
        u1 = F.conv2d(_tf_same_pad(u1, k, 1), sd[p + "u1.conva.weight"])
        out[b] = F.conv2d(_bn(sd, ns, p + "u0.bn", u1), sd[p + "u0.conv.weight"], sd[p + "u0.conv.bias"])
    return out, ns


# -- losses (utils.py:54-172), NHWC ------------------------------------------------------------
def xentropy_loss(true, pred):
    eps = 10e-8
    pred = pred / torch.sum(pred, -1, keepdim=True)
    pred = torch.clamp(pred, eps, 1.0 - eps)
    return (-torch.sum(true * torch.log(pred), -1, keepdim=True)).mean()


def dice_loss(true, pred, smooth=1e-3):
    inse = torch.sum(pred * true, (0, 1, 2))
    l, r = torch.sum(pred, (0, 1, 2)), torch.sum(true, (0, 1, 2))
    return torch.sum(1.0 - (2.0 * inse + smooth) / (l + r + smooth))


def mse_loss(true, pred):
    d = pred - true
    return (d * d).mean()


def sobel5(dtype=torch.float32):
    r = torch.arange(-2, 3, dtype=dtype)
    h, v = torch.meshgrid(r, r, indexing="ij")                                   # utils.py:135 (old default = 'ij')
    return h / (h * h + v * v + 1.0e-15), v / (h * h + v * v + 1.0e-15)


def msge_loss(true, pred, focus):
    kh, kv = sobel5(pred.dtype)

    def grad_hv(hv):
        dh = F.conv2d(hv[..., 0].unsqueeze(1), kh.view(1, 1, 5, 5), padding=2)
        dv = F.conv2d(hv[..., 1].unsqueeze(1), kv.view(1, 1, 5, 5), padding=2)
        return torch.cat([dh, dv], 1).permute(0, 2, 3, 1)

    focus = torch.stack([focus, focus], -1).type(pred.dtype)
    d = grad_hv(pred) - grad_hv(true)
    return (focus * (d * d)).sum() / (focus.sum() + 1.0e-8)


LOSS_OPTS = {"np": ("bce", "dice"), "hv": ("mse", "msge"), "tp": ("bce", "dice")}       # opt.py:47-51, weights 1


def loss_terms(logits, batch, nr_types, dtype=torch.float32, loss_opts=None):
    """logits: dict of NCHW tensors; batch: dict of tensors (np_map int64, hv_map float32, tp_map int64).
    -> (total loss, dict of named terms) exactly as run_desc.py:40-82 composes them.  dtype=float64 gives the
    high-precision reference used to measure the fp32 noise floor of the gradients.  loss_opts: the config's
    {branch: {term: weight}} table (opt.py:47-51), default all ones: `loss += weight * term`, terms tracked unweighted."""
    true_np = batch["np_map"].type(torch.int64)
    onehot_np = F.one_hot(true_np, 2).type(dtype)
    true = {"np": onehot_np, "hv": batch["hv_map"].type(dtype)}
    if nr_types is not None:
        true["tp"] = F.one_hot(batch["tp_map"].type(torch.int64), nr_types).type(dtype)
    pred = {k: v.permute(0, 2, 3, 1).contiguous() for k, v in logits.items()}
    pred["np"] = F.softmax(pred["np"], -1)
    if "tp" in pred:
        pred["tp"] = F.softmax(pred["tp"], -1)
    fn = {"bce": xentropy_loss, "dice": dice_loss, "mse": mse_loss, "msge": msge_loss}
    total, terms = 0, {}
    for b in pred:
        table = {name: 1 for name in LOSS_OPTS[b]} if loss_opts is None else loss_opts[b]
        for name, weight in table.items():
            args = [true[b], pred[b]] + ([onehot_np[..., 1]] if name == "msge" else [])
            t = fn[name](*args)
            terms["loss_%s_%s" % (b, name)] = t
            total = total + weight * t
    return total, terms


def train_step(sd, batch, mode="original", nr_types=None, freeze=False, dtype=torch.float32):
    """sd: reference-format state_dict (float32 CPU tensors); batch: numpy/tensor dict (synth_train_batch).
    -> dict(loss, terms{name: float}, grads{key: tensor | None}, new_stats{key: tensor}, logits{...})."""
    sd = {k: v.type(dtype) if v.is_floating_point() else v for k, v in sd.items()}
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k and "unpool" not in k else v)
          for k, v in sd.items()}
    batch = {k: torch.as_tensor(v) for k, v in batch.items()}
    imgs = batch["img"].type(dtype).permute(0, 3, 1, 2).contiguous()
    logits, ns = forward_train(sd, imgs, mode, freeze)
    total, terms = loss_terms(logits, batch, nr_types, dtype)
    total.backward()
    grads = {k: v.grad for k, v in sd.items() if v.is_floating_point() and v.requires_grad}
    return {"loss": float(total.detach()), "terms": {k: float(v.detach()) for k, v in terms.items()}, "grads": grads, "new_stats": ns,
            "logits": {k: v.detach() for k, v in logits.items()}}

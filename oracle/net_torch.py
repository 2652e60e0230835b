"""Plain-PyTorch fp32 restatement of the reference HoVer-Net forward and of the
`infer_step` epilogue -- TEST INFRASTRUCTURE ONLY (the floating-point oracle).

Written as one function over a reference-format `state_dict` (no nn.Module), so that
it travels to the GPU box where /root/reference does not exist.  It is pinned against
the real reference in this container by tests/test_oracle_net.py (imports
/root/reference/models/hovernet/net_desc.py when present) and by the committed
golden logits under tests/golden/net_*.npz (oracle/make_golden_net.py).

What each block follows (all paths under /root/reference/models/hovernet/):
  forward                 net_desc.py:101-145
  conv0 (+TF same pad)    net_desc.py:27-35
  residual block          net_utils.py:155-266
  dense block             net_utils.py:71-151
  TF "same" padding       net_utils.py:39-67
  nearest 2x upsample     net_utils.py:270-294
  crop_op / crop_to_shape utils.py:11-50
  infer_step epilogue     run_desc.py:185-194
"""
import torch
import torch.nn.functional as F

EPS = 1e-5
RES_UNITS = {"d0": 3, "d1": 4, "d2": 6, "d3": 3}
RES_STRIDE = {"d0": 1, "d1": 2, "d2": 2, "d3": 2}


def _bn(sd, key, x, relu=True):
    y = F.batch_norm(
        x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
        training=False, eps=EPS,
    )
    return F.relu(y) if relu else y


def _tf_same_pad(x, ksize, stride):
    # net_utils.py:52-63
    if x.shape[2] % stride == 0:
        pad = max(ksize - stride, 0)
    else:
        pad = max(ksize - (x.shape[2] % stride), 0)
    lo = pad // 2
    hi = pad - lo
    return F.pad(x, (lo, hi, lo, hi), "constant", 0)


def _crop(x, cy, cx):
    # utils.py:20-26: crop_t = c // 2, crop_b = c - crop_t
    t, l = cy // 2, cx // 2
    b, r = cy - t, cx - l
    return x[:, :, t:x.shape[2] - b, l:x.shape[3] - r]


def _upsample2x(x):
    # net_utils.py:284-294 is an exact nearest-neighbour x2 (multiplication by ones)
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def _res_block(sd, name, x, taps=None):
    stride = RES_STRIDE[name]
    shortcut = F.conv2d(x, sd[name + ".shortcut.weight"], stride=stride)  # net_utils.py:229-230
    prev = x
    for i in range(RES_UNITS[name]):
        p = "%s.units.%d." % (name, i)
        f = prev
        if i != 0:  # net_utils.py:225
            f = _bn(sd, p + "preact/bn", f)
        f = F.conv2d(f, sd[p + "conv1.weight"])
        f = _bn(sd, p + "conv1/bn", f)
        s = stride if i == 0 else 1
        f = _tf_same_pad(f, 3, s)
        f = F.conv2d(f, sd[p + "conv2.weight"], stride=s)
        f = _bn(sd, p + "conv2/bn", f)
        f = F.conv2d(f, sd[p + "conv3.weight"])
        prev = f + shortcut  # net_utils.py:263-264
        shortcut = prev
    if taps is not None:
        taps[name + ".sum"] = prev      # the block's un-normalised running sum (net_utils.py:250-266), before blk_bna
    return _bn(sd, name + ".blk_bna.bn", prev)


def _dense_block(sd, name, x, n_units):
    for i in range(n_units):
        p = "%s.units.%d." % (name, i)
        f = _bn(sd, p + "preact_bna/bn", x)
        f = F.conv2d(f, sd[p + "conv1.weight"])
        f = _bn(sd, p + "conv1/bn", f)
        f = F.conv2d(f, sd[p + "conv2.weight"], groups=4)
        x = _crop(x, x.shape[2] - f.shape[2], x.shape[3] - f.shape[3])
        x = torch.cat([x, f], dim=1)  # net_utils.py:147-148
    return _bn(sd, name + ".blk_bna.bn", x)


def forward(sd, imgs, mode="original", taps=None):
    """sd: reference-format state_dict of float32 CPU tensors; imgs: float32 NCHW in 0..255.
    Returns dict in the reference's key order (tp?, np, hv) of raw logits.
    `taps` (optional dict) receives intermediate tensors for stage-wise tests."""
    assert mode in ("original", "fast")
    with torch.no_grad():
        x = imgs / 255.0
        if mode == "fast":
            x = _tf_same_pad(x, 7, 1)
        x = F.conv2d(x, sd["conv0./.weight"])
        x = _bn(sd, "conv0.bn", x)
        d = []
        for name in ("d0", "d1", "d2", "d3"):
            x = _res_block(sd, name, x, taps)
            d.append(x)
        d[3] = F.conv2d(d[3], sd["conv_bot.weight"])
        if mode == "original":
            d[0] = _crop(d[0], 184, 184)
            d[1] = _crop(d[1], 72, 72)
        else:
            d[0] = _crop(d[0], 92, 92)
            d[1] = _crop(d[1], 36, 36)
        if taps is not None:
            taps.update(d0=d[0], d1=d[1], d2=d[2], d3=d[3])
        branches = [b for b in ("tp", "np", "hv") if ("decoder.%s.u0.conv.weight" % b) in sd]
        out = {}
        for b in branches:
            p = "decoder.%s." % b
            u3 = _upsample2x(d[3]) + d[2]
            u3 = F.conv2d(u3, sd[p + "u3.conva.weight"])
            u3 = _dense_block(sd, p + "u3.dense", u3, 8)
            u3 = F.conv2d(u3, sd[p + "u3.convf.weight"])
            u2 = _upsample2x(u3) + d[1]
            u2 = F.conv2d(u2, sd[p + "u2.conva.weight"])
            u2 = _dense_block(sd, p + "u2.dense", u2, 4)
            u2 = F.conv2d(u2, sd[p + "u2.convf.weight"])
            u1 = _upsample2x(u2) + d[0]
            k = sd[p + "u1.conva.weight"].shape[2]
            u1 = F.conv2d(_tf_same_pad(u1, k, 1), sd[p + "u1.conva.weight"])
            u0 = _bn(sd, p + "u0.bn", u1)
            if taps is not None:
                taps[b + ".u3"] = u3
                taps[b + ".u2"] = u2
                taps[b + ".u1"] = u0
            out[b] = F.conv2d(u0, sd[p + "u0.conv.weight"], sd[p + "u0.conv.bias"])
        return out


def infer_epilogue(out):
    """run_desc.py:185-194: logits dict -> float32 [N,h,w,3|4] = [type?, p_nuc, h, v]."""
    with torch.no_grad():
        chans = []
        if "tp" in out:
            t = F.softmax(out["tp"].permute(0, 2, 3, 1), dim=-1)
            chans.append(torch.argmax(t, dim=-1, keepdim=True).type(torch.float32))
        chans.append(F.softmax(out["np"].permute(0, 2, 3, 1), dim=-1)[..., 1:])
        chans.append(out["hv"].permute(0, 2, 3, 1))
        return torch.cat(chans, -1).contiguous()

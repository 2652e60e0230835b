"""Table-driven description of the HoVer-Net parameter set and geometry (no torch).

Everything the host side needs to know about the network is derived here from two
arguments, `mode` ('original' | 'fast') and `nr_types` (None | int):

* `param_table`  -- the exact checkpoint key set / shapes of the reference
  (`/root/reference/models/hovernet/net_desc.py:17-99`, canonical list
  `/root/reference/variables_tf2pytorch.csv` + `num_batches_tracked`), which is part of
  the drop-in contract (`infer/base.py:65-68` loads with `strict=True`).
* `geometry`     -- spatial sizes / crops of SURVEY.md Appendix C
  (`net_desc.py:124-129`, `run_infer.py:145-150`).
"""
from collections import OrderedDict

RES_BLOCKS = (  # name, in_ch, (c1, c2, c3), units, stride       net_desc.py:36-39
    ("d0", 64, (64, 64, 256), 3, 1),
    ("d1", 256, (128, 128, 512), 4, 2),
    ("d2", 512, (256, 256, 1024), 6, 2),
    ("d3", 1024, (512, 512, 2048), 3, 2),
)
DENSE_GROWTH = 32      # net_desc.py:46,53  unit_ch = [128, 32]
DENSE_MID = 128
DENSE_GROUPS = 4
BN_EPS = 1e-5


def branch_names(nr_types):
    return ("np", "hv") if nr_types is None else ("tp", "np", "hv")  # net_desc.py:77-97


def branch_out_ch(branch, nr_types):
    return nr_types if branch == "tp" else 2


def decoder_ksize(mode):
    return 5 if mode == "original" else 3  # net_desc.py:76


def _bn(t, key, ch):
    t[key + ".weight"] = ("bn_w", (ch,))
    t[key + ".bias"] = ("bn_b", (ch,))
    t[key + ".running_mean"] = ("bn_rm", (ch,))
    t[key + ".running_var"] = ("bn_rv", (ch,))
    t[key + ".num_batches_tracked"] = ("bn_nbt", ())


def param_table(mode="original", nr_types=None, input_ch=3):
    """OrderedDict key -> (kind, shape) in the reference's state_dict order.
    kind in {conv, bias, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, ones}."""
    assert mode in ("original", "fast"), "Unknown mode `%s` for HoVerNet. Only support `original` or `fast`." % mode
    t = OrderedDict()
    t["conv0./.weight"] = ("conv", (64, input_ch, 7, 7))
    _bn(t, "conv0.bn", 64)
    for name, in_ch, (c1, c2, c3), units, _stride in RES_BLOCKS:
        cin = in_ch
        for i in range(units):
            p = "%s.units.%d." % (name, i)
            if i != 0:
                _bn(t, p + "preact/bn", cin)
            t[p + "conv1.weight"] = ("conv", (c1, cin, 1, 1))
            _bn(t, p + "conv1/bn", c1)
            t[p + "conv2.weight"] = ("conv", (c2, c1, 3, 3))
            _bn(t, p + "conv2/bn", c2)
            t[p + "conv3.weight"] = ("conv", (c3, c2, 1, 1))
            cin = c3
        t[name + ".shortcut.weight"] = ("conv", (c3, in_ch, 1, 1))
        _bn(t, name + ".blk_bna.bn", c3)
    t["conv_bot.weight"] = ("conv", (1024, 2048, 1, 1))
    k = decoder_ksize(mode)
    for b in branch_names(nr_types):
        for uname, cin, cmid, units in (("u3", 1024, 256, 8), ("u2", 512, 128, 4)):
            p = "decoder.%s.%s." % (b, uname)
            t[p + "conva.weight"] = ("conv", (cmid, cin, k, k))
            c = cmid
            for i in range(units):
                q = p + "dense.units.%d." % i
                _bn(t, q + "preact_bna/bn", c)
                t[q + "conv1.weight"] = ("conv", (DENSE_MID, c, 1, 1))
                _bn(t, q + "conv1/bn", DENSE_MID)
                t[q + "conv2.weight"] = ("conv", (DENSE_GROWTH, DENSE_MID // DENSE_GROUPS, k, k))
                c += DENSE_GROWTH
            _bn(t, p + "dense.blk_bna.bn", c)
            t[p + "convf.weight"] = ("conv", (c, c, 1, 1))
        p = "decoder.%s." % b
        t[p + "u1.conva.weight"] = ("conv", (64, 256, k, k))
        _bn(t, p + "u0.bn", 64)
        t[p + "u0.conv.weight"] = ("conv", (branch_out_ch(b, nr_types), 64, 1, 1))
        t[p + "u0.conv.bias"] = ("bias", (branch_out_ch(b, nr_types),))
    t["upsample2x.unpool_mat"] = ("ones", (2, 2))
    return t


def geometry(mode="original"):
    """Spatial sizes per stage."""
    k = decoder_ksize(mode)
    if mode == "original":
        g = dict(inp=270, conv0_pad=0, d=(264, 132, 66, 33), crop0=92, crop1=36)
    else:
        g = dict(inp=256, conv0_pad=3, d=(256, 128, 64, 32), crop0=46, crop1=18)
    g["k"] = k
    s3 = g["d"][2] - (k - 1)                 # u3.conva valid
    g["u3_cat"] = s3
    g["u3_out"] = s3 - 8 * (k - 1)
    s2 = 2 * g["u3_out"] - (k - 1)           # u2.conva valid
    g["u2_cat"] = s2
    g["u2_out"] = s2 - 4 * (k - 1)
    g["out"] = 2 * g["u2_out"]               # u1 same-pad, u0 1x1
    assert g["d"][1] - 2 * g["crop1"] == 2 * g["u3_out"]
    assert g["d"][0] - 2 * g["crop0"] == g["out"]
    return g

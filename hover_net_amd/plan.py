"""Lowering of a HoVer-Net checkpoint to the fused-kernel launch plan (host logic, numpy only).

The reference executes ~360 separate torch ops per forward
(/root/reference/models/hovernet/net_desc.py:101-145).  Here the same function is
lowered ONCE, at checkpoint-load time, to ~150 launches of five kernel kinds over
channels-last (NHWC) fp32 activations that stay resident in HBM:

  CONV0    uint8 NHWC image -> 7x7 conv (1/255 and BN folded into the taps) + ReLU
  CONV     implicit-GEMM conv on fp32 MFMA with
             prologue : relu(x*ps + pb) per input channel   (pre-activation BN that
                        cannot be folded: its input is a residual sum / a concat)
             epilogue : + bias (folded BN shift), ReLU, + residual, then optional
                        relu(y*qs + qb) per output channel   (block-closing BN-ReLU)
             strided input / output views, so dense-block concats and centre crops
             are address arithmetic, never copies
  UPADD    nearest 2x upsample + (cropped) skip add
  WINO_IN / WINO_OUT  input / output transforms of the Winograd F(2x2,5x5) form of the 5x5 decoder convs
           (51 % of the network's FLOPs): 36 multiplications per 2x2 outputs instead of 100; the 36
           independent [tiles x cin] . [cin x cout] products run as ONE batched launch of the CONV kernel
  CHAIN    a residual unit's conv3 (+ residual / fused shortcut, + block-closing BN-ReLU) and the NEXT unit's pre-activation +
           conv1 in one launch (`fuse_chains`): the 1x1 -> 1x1 seam needs no neighbour pixels, so conv1 runs on the sum while it
           is still in LDS -- one read of the widest tensor of the block and one launch less per unit
  HEAD     1x1 conv to the 2..6 logits (+bias), written NCHW (the forward() contract)
  PREDMAP  infer_step's softmax / argmax / concat (run_desc.py:185-194)

A plan is pure data (`Plan.ops`, numpy weights, symbolic buffers) so it can be
interpreted by any executor: the product binds it to device memory and hands it to
libhvn_hip.so (`hover_net_amd.lib`); the tests interpret the same plan with torch CPU ops
to prove the lowering itself (folding, crops, concat offsets) against the oracle.
"""
from dataclasses import dataclass, field

import numpy as np

from . import arch

OP_CONV0, OP_CONV, OP_UPADD, OP_HEAD, OP_PREDMAP, OP_WINO_IN, OP_WINO_OUT, OP_CHAIN = 1, 2, 3, 4, 5, 6, 7, 8


@dataclass
class Buf:
    name: str
    h: int
    w: int
    c: int
    dtype: str = "f32"          # per-sample element count = h*w*c
    offset: int = -1            # per-sample float offset inside the arena, set by Plan.pack()
    first: int = 10 ** 9
    last: int = -1

    @property
    def size(self):
        return self.h * self.w * self.c


@dataclass
class View:
    buf: Buf
    y0: int = 0
    x0: int = 0
    h: int = -1
    w: int = -1
    c0: int = 0
    c: int = -1

    def __post_init__(self):
        if self.h < 0:
            self.h = self.buf.h - self.y0
        if self.w < 0:
            self.w = self.buf.w - self.x0
        if self.c < 0:
            self.c = self.buf.c - self.c0
        assert 0 <= self.y0 and self.y0 + self.h <= self.buf.h
        assert 0 <= self.x0 and self.x0 + self.w <= self.buf.w
        assert 0 <= self.c0 and self.c0 + self.c <= self.buf.c

    def crop(self, m):
        return View(self.buf, self.y0 + m, self.x0 + m, self.h - 2 * m, self.w - 2 * m, self.c0, self.c)

    def chans(self, c0, c):
        return View(self.buf, self.y0, self.x0, self.h, self.w, self.c0 + c0, c)


@dataclass
class Op:
    kind: int
    name: str
    x: View = None
    y: View = None
    res: View = None
    w: np.ndarray = None        # CONV: [cout_pad, cin/32, kh*kw, 32]; CONV0: [7,7,3,64]; HEAD: [cout, 64]
    bias: np.ndarray = None
    pre: tuple = None           # (scale[cin], shift[cin])
    post: tuple = None          # (scale[cout], shift[cout])
    kh: int = 1
    kw: int = 1
    stride: int = 1
    pad_t: int = 0
    pad_l: int = 0
    relu: int = 0
    cout: int = 0
    tile_n: int = 0
    extra: dict = field(default_factory=dict)

    def flops(self):
        if "algo_flops" in self.extra:        # Winograd GEMM: credited with the direct convolution's work
            return self.extra["algo_flops"]
        if self.kind == OP_CONV:
            return 2.0 * self.y.h * self.y.w * self.cout * self.kh * self.kw * self.extra.get("cin_real", self.x.c)
        if self.kind == OP_CHAIN:
            return 2.0 * self.y.h * self.y.w * (self.cout * self.extra["cin_real"] + self.extra["cout2"] * self.cout)
        if self.kind == OP_CONV0:
            return 2.0 * self.y.h * self.y.w * 64 * 147
        if self.kind == OP_HEAD:
            return 2.0 * self.y.h * self.y.w * self.cout * 64
        return 0.0


def _f64(sd, key):
    v = sd[key]
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, np.float64)


def _bn_affine(sd, key):
    """BatchNorm2d(eval) as y = x*s + b   (torch semantics, eps 1e-5)."""
    s = _f64(sd, key + ".weight") / np.sqrt(_f64(sd, key + ".running_var") + arch.BN_EPS)
    b = _f64(sd, key + ".bias") - _f64(sd, key + ".running_mean") * s
    return s, b


def _tf_same(size, k, s):
    """TFSamepaddingLayer (net_utils.py:52-63) -> (pad_lo, pad_hi)."""
    pad = max(k - s, 0) if size % s == 0 else max(k - (size % s), 0)
    return pad // 2, pad - pad // 2


def _tile_n(cout):
    return 128 if cout >= 128 else (64 if cout >= 64 else 32)


# Transform-domain weight packings of the last plans built in this process, keyed by a hash of the (BN-folded, float64) weights: the float64
# GEMM with kron(G, G) and the packing are ~85 % of `build_plan`'s time (3.4 of 4.0 s for a cfg-2 plan), and a process that rebuilds an
# engine for the same checkpoint -- another batch size, dtype, lowering or launch schedule; every second test of the GPU suite -- gets the
# same arrays again.  Entries are read-only; least recently used ones go once the cache holds more than HVN_WINO_CACHE_MB (default 2048).
_WINO_PACKED = {}
_WINO_PACKED_BYTES = [0]


def _wino_packed(wt, m, n2, cout, cin, cout_pad):
    """U = G g G^T of every (cout, cin) filter, packed [n2, cout_pad, cin / 32, 1, 32] fp32 (see `conv_winograd`)."""
    import os

    from . import winograd as WG
    try:
        import xxhash
        digest = xxhash.xxh3_128_hexdigest(np.ascontiguousarray(wt, np.float64))
    except ImportError:                                    # pragma: no cover
        import hashlib
        digest = hashlib.blake2b(np.ascontiguousarray(wt, np.float64), digest_size=16).hexdigest()
    key = (digest, tuple(wt.shape), m, cout_pad)
    hit = _WINO_PACKED.pop(key, None)
    if hit is None:
        u = WG.transform_weights(wt, m)                                       # [n2, cout, cin] float64
        hit = np.zeros((n2, cout_pad, cin // 32, 1, 32), np.float32)
        hit[:, :cout] = u.reshape(n2, cout, cin // 32, 1, 32)
        _WINO_PACKED_BYTES[0] += hit.nbytes
    _WINO_PACKED[key] = hit                                                   # (re)inserted last = most recently used
    limit = int(os.environ.get("HVN_WINO_CACHE_MB", "2048")) << 20
    while _WINO_PACKED_BYTES[0] > limit and len(_WINO_PACKED) > 1:
        old_key = next(iter(_WINO_PACKED))
        _WINO_PACKED_BYTES[0] -= _WINO_PACKED.pop(old_key).nbytes
    return hit


def _pack_conv(wt, out_scale=None, groups=1):
    """[cout, cin/groups, kh, kw] (torch) -> [cout_pad, cin/32, kh*kw, 32] fp32: BN scale folded in
    float64, grouped convs expanded to block-diagonal dense, and the reduction index ordered
    (channel chunk of 32, tap, channel) -- the kernel walks all taps of one 32-channel slab before the
    next slab, so the shifted input windows of the taps are re-read from L1/L2 instead of HBM."""
    cout, cin_g, kh, kw = wt.shape
    if out_scale is not None:
        wt = wt * out_scale[:, None, None, None]
    cin = cin_g * groups
    assert cin % 32 == 0
    tn = _tile_n(cout)
    cout_pad = (cout + tn - 1) // tn * tn
    dense = np.zeros((cout_pad, kh * kw, cin), np.float64)
    og = cout // groups
    for g in range(groups):
        blk = wt[g * og:(g + 1) * og]                                  # [og, cin_g, kh, kw]
        dense[g * og:(g + 1) * og, :, g * cin_g:(g + 1) * cin_g] = blk.transpose(0, 2, 3, 1).reshape(og, kh * kw, cin_g)
    packed = dense.reshape(cout_pad, kh * kw, cin // 32, 32).transpose(0, 2, 1, 3)
    return np.ascontiguousarray(packed, np.float32), tn


def unpack_conv(w, cout):
    """Inverse of the packing for interpreters: -> [cout, cin, taps] float32."""
    cout_pad, chunks, taps, _ = w.shape
    return np.ascontiguousarray(w[:cout].transpose(0, 1, 3, 2).reshape(cout, chunks * 32, taps))


class Plan:
    def __init__(self, mode, nr_types):
        self.mode = mode
        self.nr_types = nr_types
        self.geo = arch.geometry(mode)
        self.ops = []
        self.bufs = []
        self.logits = {}        # branch -> Buf (NCHW, c=out_ch)
        self.pred_map = None    # Buf [h, w, 3|4]
        self.image = Buf("image", self.geo["inp"], self.geo["inp"], 3, "u8")
        self.arena_per_sample = 0
        self.lanes = None

    # -- construction helpers -------------------------------------------------
    def buf(self, name, h, w, c):
        b = Buf(name, h, w, c)
        self.bufs.append(b)
        return b

    def add(self, op):
        i = len(self.ops)
        for v in (op.x, op.y, op.res, op.extra.get("y2")):
            if v is not None and v.buf.dtype == "f32":
                v.buf.first = min(v.buf.first, i)
                v.buf.last = max(v.buf.last, i)
        for b in op.extra.get("reads", ()):
            b.first = min(b.first, i)
            b.last = max(b.last, i)
        self.ops.append(op)
        return op

    def conv(self, name, x, y, wt, *, stride=1, pad=(0, 0), bn=None, relu=0, pre=None, res=None, post=None, groups=1, bias=None,
             x2=None, wt2=None, stride2=1):
        """x2 / wt2 / stride2: a second 1x1 input whose channels are appended to the reduction (fused shortcut)."""
        s = b = None
        if bn is not None:
            s, b = bn
        if x2 is not None:
            assert wt.shape[2:] == (1, 1) and wt2.shape[2:] == (1, 1) and stride == 1 and groups == 1 and pre is None
            wt = np.concatenate([wt, wt2], axis=1)
        w, tn = _pack_conv(wt, s, groups)
        cout, _cin_g, kh, kw = wt.shape
        assert x.c + (x2.c if x2 is not None else 0) == w.shape[1] * 32 and y.c == cout, (name, x.c, w.shape, y.c, cout)
        assert y.h == (x.h + pad[0] + pad[1] - kh) // stride + 1, (name, x.h, y.h)
        if bias is not None:
            b = bias if b is None else b + bias
        op = Op(OP_CONV, name, x=x, y=y, res=res, w=w, kh=kh, kw=kw, stride=stride, pad_t=pad[0], pad_l=pad[0],
                relu=relu, cout=cout, tile_n=tn)
        op.bias = None if b is None else np.asarray(b, np.float32)
        op.pre = None if pre is None else (np.asarray(pre[0], np.float32), np.asarray(pre[1], np.float32))
        op.post = None if post is None else (np.asarray(post[0], np.float32), np.asarray(post[1], np.float32))
        op.extra["cin_real"] = (x.c + (x2.c if x2 is not None else 0)) // groups
        op.extra["groups"] = groups
        if x2 is not None:
            assert y.h == (x2.h - 1) // stride2 + 1
            op.extra["x2"], op.extra["stride2"] = x2, stride2
            if x2.buf.dtype == "f32":
                op.extra["reads"] = [x2.buf]
        return self.add(op)

    def conv_winograd(self, name, x, y, wt, *, pad=(0, 0), bn=None, relu=0, share_in=False, m=4):
        """r x r (5x5 or 3x3) stride-1 conv as Winograd F(m x m, r x r): WINO_IN (shared between convs that read the
        same view with the same padding) -> batched CONV over the (m+r-1)^2 transform positions -> WINO_OUT (+bias,
        ReLU).  Output extents that are not a multiple of m get a partial last tile (its surplus outputs are not
        written)."""
        from . import winograd as WG

        cout, cin, kh, kw = wt.shape
        assert kh == kw and (kh, m) in ((5, 2), (5, 4), (5, 6), (3, 4), (3, 6)) and x.c == cin and y.c == cout
        r = kh
        assert y.h == x.h + pad[0] + pad[1] - (r - 1)
        if m == 6 and -(-y.h // 6) * -(-y.w // 6) < 64:
            # Few tiles per sample (d3: 33^2 | 32^2 -> 36): a 128-row tile of the transform-domain product then reaches 4 samples
            # ahead, and with the arena's sample stride (0.5 .. 0.8 GB) that is beyond the kernels' 32-bit offsets (the launchers
            # refuse it since round 4; before, 'fast' mode at batch >= 8 silently read zeros there).  F(4x4, r x r) has 64 .. 81.
            m = 4
        at, _g, bt = WG.mats(m, r)
        n2 = (m + r - 1) ** 2
        s = b = None
        if bn is not None:
            s, b = bn
            wt = wt * s[:, None, None, None]
        ty, tx = -(-y.h // m), -(-y.w // m)
        t1 = ty * tx
        key = (id(x.buf), x.y0, x.x0, x.h, x.w, x.c0, x.c, pad[0], m, r)
        cache = self.__dict__.setdefault("_wino_in", {})
        if key not in cache:
            vbuf = self.buf(name + ".V", n2, t1, cin)
            op = Op(OP_WINO_IN, name + ".wino_in", x=x, y=View(vbuf), w=np.ascontiguousarray(bt, np.float32),
                    pad_t=pad[0], pad_l=pad[0], extra={"tiles": (ty, tx), "shared": share_in, "m": m, "r": r})
            self.add(op)
            cache[key] = vbuf
        vbuf = cache[key]
        mbuf = self.buf(name + ".M", n2, t1, cout)
        tn = _tile_n(cout)
        cout_pad = (cout + tn - 1) // tn * tn
        packed = _wino_packed(wt, m, n2, cout, cin, cout_pad)
        g = Op(OP_CONV, name + ".wino_gemm", x=View(vbuf, 0, 0, 1, t1), y=View(mbuf, 0, 0, 1, t1), w=packed, cout=cout, tile_n=tn)
        g.extra.update(nbatch=n2, batch_strides=(t1 * cin, cout_pad * cin, t1 * cout), cin_real=cin, groups=1,
                       algo_flops=2.0 * y.h * y.w * cout * cin * r * r, exec_flops=2.0 * n2 * t1 * cout * cin)
        self.add(g)
        for bb in (vbuf, mbuf):   # the batched launch touches every row of V and M
            bb.last = max(bb.last, len(self.ops) - 1)
        o = Op(OP_WINO_OUT, name + ".wino_out", x=View(mbuf), y=y, w=np.ascontiguousarray(at, np.float32),
               bias=None if b is None else np.asarray(b, np.float32), relu=relu, cout=cout, extra={"tiles": (ty, tx), "m": m, "r": r})
        return self.add(o)

    def mark_x3(self, terms, d1=True, chain="d0"):
        """Which CONV launches run on csrc/hvn_conv_x3.hip (fp32 in / out, products on the bf16 matrix pipe from exact bf16x3 splits,
        `terms` = 9 | 6 partial products per product): every dense (ungrouped) conv with a 128- or 64-wide column tile EXCEPT the 1x1
        convs of d0 (+ d1's first conv1, chained to d0's last conv3) -- HBM-bound, they stay chained pairs on the fp32 pipe
        (`fuse_chains`, which leaves marked ops alone).  d1's 1x1 convs (d1=True) run here unchained: measured faster than their
        chains (655 -> 674 tiles/s).  The rule is by layer, not by what the chain pass did -- the chained and the unchained lowering
        of a checkpoint stay bit-identical -- and static, not a timing: a tile's bits must not depend on the batch it is run in."""
        import re

        assert terms in (9, 6), terms
        for op in self.ops:
            if op.kind != OP_CONV or int(op.extra.get("groups", 1)) != 1 or op.tile_n not in (128, 64):
                continue
            seam = False
            if re.match(r"^d0\.units\.\d+\.conv[13]$", op.name) or op.name == "d1.units.0.conv1":
                # round 5: d0's seams (conv3 -> next conv1, + d0's last conv3 -> d1's first conv1) run CHAINED on the bf16 pipe
                # (csrc/hvn_conv_chain_x3.hip; chain = "d0" | "d0d1"); chain = "" keeps them chained on the fp32 pipe (rounds 3-4)
                if not chain or op.name == "d0.units.0.conv1":
                    continue
                seam = True
            if re.match(r"^d1\.units\.\d+\.conv[13]$", op.name) and op.name != "d1.units.0.conv1":
                if not d1:
                    continue            # d1=True (default; HVN_X3_D1=0 chains them on the fp32 pipe instead): measured 655 -> 674 tiles/s
                seam = chain == "d0d1"
            op.extra["x3"] = terms
            if seam:
                op.extra["x3_chain"] = True     # `fuse_chains` may merge two such ops into one OP_CHAIN on the bf16 pipe

    def reindex(self):
        """Recompute every buffer's live interval from the op list (after a pass that merged / removed ops)."""
        ops, self.ops = self.ops, []
        for b in self.bufs:
            b.first, b.last = 10 ** 9, -1
        for op in ops:
            self.add(op)

    def fuse_chains(self, max_n2=128, accept=None):
        """Merge each residual unit's closing 1x1 conv (conv3: + residual or fused shortcut, optional block-closing BN-ReLU)
        with the 1x1 conv that directly consumes its output (the next unit's conv1, pre-activation included) into one
        OP_CHAIN launch (csrc/hvn_conv_chain.hip; net_utils.py:250-266).  Both are per-pixel, so the second runs on the
        first's output while it is still on chip; results are bit-identical to the two launches."""
        out, i, ops = [], 0, self.ops
        while i < len(ops):
            a = ops[i]
            b = ops[i + 1] if i + 1 < len(ops) else None

            def plain1x1(o):
                return (o is not None and o.kind == OP_CONV and o.kh == 1 and o.kw == 1 and o.stride == 1 and o.pad_t == 0 and
                        not o.extra.get("nbatch") and o.extra.get("groups", 1) == 1 and (not o.extra.get("x3") or o.extra.get("x3_chain")))

            ok = (plain1x1(a) and plain1x1(b) and int(a.extra.get("x3", 0)) == int(b.extra.get("x3", 0)) and (a.res is not None or a.extra.get("x2") is not None) and a.pre is None and
                  a.bias is None and not a.relu and a.cout % 64 == 0 and a.x.c + (a.extra["x2"].c if a.extra.get("x2") is not None else 0) >= 64 and
                  b.res is None and b.extra.get("x2") is None and b.post is None and b.relu == 1 and
                  b.cout in (64, 128) and b.cout <= max_n2 and
                  (b.x.buf is a.y.buf and (b.x.y0, b.x.x0, b.x.h, b.x.w, b.x.c0, b.x.c) == (a.y.y0, a.y.x0, a.y.h, a.y.w, a.y.c0, a.y.c)) and
                  (b.y.h, b.y.w) == (a.y.h, a.y.w) and
                  (a.res is None or (a.res.buf.w, a.res.buf.c, a.res.h, a.res.w, a.res.c) == (a.y.buf.w, a.y.buf.c, a.y.h, a.y.w, a.y.c)))
            if ok and accept is not None and not accept(a, b):
                ok = False
            if not ok:
                out.append(a)
                i += 1
                continue
            op = Op(OP_CHAIN, a.name + "+" + b.name.split(".", 1)[1] if b.name.split(".")[0] == a.name.split(".")[0] else a.name + "+" + b.name,
                    x=a.x, y=a.y, res=a.res, w=a.w, post=a.post, pre=b.pre, cout=a.cout, tile_n=0)
            op.extra.update(cin_real=a.extra["cin_real"], groups=1, cout2=b.cout, w2=b.w, bias2=b.bias, y2=b.y, parts=(a, b))
            if a.extra.get("x3"):
                op.extra["x3"] = int(a.extra["x3"])      # both GEMMs on the bf16 pipe (csrc/hvn_conv_chain_x3.hip)
            if a.extra.get("x2") is not None:
                op.extra.update(x2=a.extra["x2"], stride2=a.extra["stride2"], reads=a.extra.get("reads", []))
            out.append(op)
            i += 2
        self.ops = out
        self.reindex()

    def fuse_upadd_into_winograd(self):
        """An UPADD (UpSample2x + skip add, net_utils.py:284-294 / net_desc.py:133-143) whose output is read ONLY by Winograd input
        transforms disappears into them: WINO_IN then forms nearest2x(lo) + skip on the fly (`res` = the half-resolution view, `x` =
        the skip) -- the up-sampled tensor is never written or re-read and one launch per decoder stage goes away.  The sum is the one
        `hvn_upadd` forms, so the transform-domain tensor has the same bits."""
        users = {}
        for op in self.ops:
            for v in (op.x, op.res, op.extra.get("x2")):
                if v is not None:
                    users.setdefault(id(v.buf), []).append(op)
        keep = []
        for op in self.ops:
            if op.kind == OP_UPADD:
                readers = users.get(id(op.y.buf), [])
                whole = lambda v: (v.y0, v.x0, v.h, v.w, v.c0, v.c) == (0, 0, op.y.buf.h, op.y.buf.w, 0, op.y.buf.c)   # noqa: E731
                if readers and all(r.kind == OP_WINO_IN and r.res is None and r.x.buf is op.y.buf and whole(r.x) for r in readers) and whole(op.y):
                    for r in readers:
                        r.x, r.res = op.res, op.x            # x = the skip (full resolution), res = the tensor to up-sample
                        r.extra["upadd"] = op.name
                    continue
            keep.append(op)
        self.ops = keep
        self.reindex()

    # -- memory planning --------------------------------------------------------
    def pack(self, align=64):
        """Greedy interval packing of the per-sample activation arena (floats)."""
        live = [b for b in self.bufs if b.last >= 0]
        placed = []
        for b in sorted(live, key=lambda t: -t.size):
            sz = (b.size + align - 1) // align * align
            busy = sorted((p.offset, p.offset + (p.size + align - 1) // align * align) for p in placed
                          if not (p.last < b.first or b.last < p.first))
            off = 0
            for lo, hi in busy:
                if off + sz <= lo:
                    break
                off = max(off, hi)
            b.offset = off
            placed.append(b)
        self.arena_per_sample = max((p.offset + (p.size + align - 1) // align * align) for p in placed)
        return self.arena_per_sample

    def total_flops(self):
        return sum(o.flops() for o in self.ops)


LOWERINGS = ("default", "conservative")


def build_plan(sd, mode="original", nr_types=None, with_predmap=True, winograd=None, chain=None, x3=None, lowering="default"):
    """sd: reference-format state_dict (torch tensors or numpy arrays).
    winograd: output tile m (2 or 4) of the Winograd F(m x m, 5x5) form of the 5x5 decoder convs; 0 / False = direct
    convolution; default 4 (env HVN_WINOGRAD).
    chain: fuse conv3 -> next conv1 seams of the encoder into OP_CHAIN launches (`Plan.fuse_chains`); default on
    (env HVN_CHAIN=0 turns it off, HVN_CHAIN_MAXN2 = 64 | 128 bounds the second conv's width); fp32 only.
    x3: 0 = every fp32 conv on the fp32 matrix pipe; 9 | 6 = the MFMA-bound conv launches form their products on the bf16 matrix pipe
    from exact three-way bf16 splits of the fp32 operands (csrc/hvn_conv_x3.hip, `Plan.mark_x3`); default env HVN_X3; fp32 only."""
    import os
    if lowering not in LOWERINGS:
        raise ValueError("lowering must be one of %s, got %r" % (LOWERINGS, lowering))
    conservative = lowering == "conservative"
    # lowering="conservative" (HoVerNet.lowering; no environment involved): for checkpoints whose activations run hotter than the ones the
    # default was qualified on (tests/test_gpu_trained_like.py: logits within 1e-3 of the fp32 oracle up to max |activation| ~ 10^2) --
    # F(4x4, .) Winograd tiles everywhere (the smallest transform constants) and all NINE partial products of the bf16x3 convolution
    # (the fp32 dot product in another summation order, nothing dropped).  ~15 % slower.
    if chain is None:
        chain = os.environ.get("HVN_CHAIN", "1") != "0"
    if winograd is None:
        winograd = 4 if conservative else int(os.environ.get("HVN_WINOGRAD", "4"))      # output tile m of F(m x m, 5x5); 0 = direct conv
    wino_m = 4 if winograd is True else int(winograd)
    # per decoder stage override of the 5x5 output tile, e.g. HVN_WINOGRAD_STAGES="u3:6" (F(6x6,5x5) for u3.conva only, the rest as
    # HVN_WINOGRAD says): F(6x6,5x5) costs ~8x the fp32 error of F(4x4,5x5) (tests/test_gpu_trained_like.py), which stage carries it matters
    # Default since round 4: u3 (1024 -> 256 @62^2, the largest product) as F(6x6,5x5): 1.95e-4 on the trained-like checkpoint against
    # 1.70e-4 without and 1.95e-4 for direct convolutions; u2 (4.0e-4) and u1 (1.15e-3: it feeds the logits directly) keep F(4x4,5x5).
    wino_stage = {kv.split(":")[0]: int(kv.split(":")[1]) for kv in os.environ.get("HVN_WINOGRAD_STAGES", "u3:6").split(",") if ":" in kv}
    if not winograd or conservative:
        wino_stage = {}
    wino3 = int(os.environ.get("HVN_WINOGRAD3", "128"))      # minimum channel count for the Winograd form of the encoder's 3x3 convs; 0 = off
    wino3_m = int(os.environ.get("HVN_WINOGRAD3_M", "6"))     # its output tile: F(6x6,3x3) (default since round 4: same logit error, 1.78 instead of 2.25 multiplies per output) or F(4x4,3x3)
    if conservative:
        wino3_m = 4
    P = Plan(mode, nr_types)
    g = P.geo
    k = g["k"]
    W = lambda key: _f64(sd, key)          # noqa: E731
    BN = lambda key: _bn_affine(sd, key)   # noqa: E731

    # conv0: /255 and BN folded -> [7,7,3,64] taps, bias, ReLU   (net_desc.py:27-35,103)
    s, b = BN("conv0.bn")
    w0 = W("conv0./.weight") * (s / 255.0)[:, None, None, None]
    d_sz = g["d"]
    x = P.buf("conv0", d_sz[0], d_sz[0], 64)
    op = Op(OP_CONV0, "conv0", x=View(P.image), y=View(x), w=np.ascontiguousarray(w0.transpose(2, 3, 1, 0), np.float32),
            bias=np.asarray(b, np.float32), kh=7, kw=7, pad_t=g["conv0_pad"], pad_l=g["conv0_pad"], relu=1, cout=64)
    P.add(op)
    x = View(x)

    # encoder: four pre-activation residual blocks   (net_utils.py:155-266)
    skips = []
    for bi, (name, in_ch, (c1, c2, c3), units, stride) in enumerate(arch.RES_BLOCKS):
        hi, ho = x.h, d_sz[bi]
        acc = View(P.buf(name + ".sum", ho, ho, c3))
        block_in = x
        cur = x
        for i in range(units):
            p = "%s.units.%d." % (name, i)
            st = stride if i == 0 else 1
            t1 = View(P.buf(p + "t1", cur.h, cur.w, c1))
            P.conv(p + "conv1", cur, t1, W(p + "conv1.weight"), bn=BN(p + "conv1/bn"), relu=1,
                   pre=None if i == 0 else BN(p + "preact/bn"))
            t2 = View(P.buf(p + "t2", ho, ho, c2))
            if st == 1 and winograd and wino3 and c1 >= wino3:
                # stride-1 3x3 with K >= 128 (d1, d2, d3): Winograd F(4x4,3x3), 2.25 instead of 9 multiplies per output;
                # below that the transform-domain tensors make the layer HBM-bound (d0: K = 64); measured 453 (off) / 484 (K >= 256) / 497 (K >= 128) tiles/s
                P.conv_winograd(p + "conv2", t1, t2, W(p + "conv2.weight"), pad=_tf_same(t1.h, 3, 1), bn=BN(p + "conv2/bn"), relu=1, m=wino3_m)
            else:
                P.conv(p + "conv2", t1, t2, W(p + "conv2.weight"), stride=st, pad=_tf_same(t1.h, 3, st),
                       bn=BN(p + "conv2/bn"), relu=1)
            last = i == units - 1
            out = View(P.buf(name + ".out", ho, ho, c3)) if last else acc
            if i == 0:
                # unit 0: the strided 1x1 shortcut (net_utils.py:229-230) is a second input of the same GEMM
                # (K = c2 + in_ch): its result is never written to / re-read from HBM as a residual
                P.conv(p + "conv3", t2, out, W(p + "conv3.weight"), x2=block_in, wt2=W(name + ".shortcut.weight"), stride2=stride,
                       post=BN(name + ".blk_bna.bn") if last else None)
            else:
                P.conv(p + "conv3", t2, out, W(p + "conv3.weight"), res=acc,
                       post=BN(name + ".blk_bna.bn") if last else None)
            cur = acc
        x = out
        skips.append(out)
        del hi
    d3 = View(P.buf("conv_bot", d_sz[3], d_sz[3], 1024))
    P.conv("conv_bot", skips[3], d3, W("conv_bot.weight"))

    # decoder: the u3 input is branch-independent (net_desc.py:133) -> computed once
    u3in = View(P.buf("u3.in", d_sz[2], d_sz[2], 1024))
    P.add(Op(OP_UPADD, "u3.upadd", x=d3, res=skips[2], y=u3in))
    d1c = skips[1].crop(g["crop1"])
    d0c = skips[0].crop(g["crop0"])
    cm = (k - 1) // 2
    heads = {}
    for br in arch.branch_names(nr_types):
        pb = "decoder.%s." % br
        cur_in = u3in
        for uname, cmid, units, cat_sz, skip in (("u3", 256, 8, g["u3_cat"], d1c), ("u2", 128, 4, g["u2_cat"], d0c)):
            p = pb + uname + "."
            ctot = cmid + units * arch.DENSE_GROWTH
            cat = View(P.buf(p + "cat", cat_sz, cat_sz, ctot))
            if winograd and (k == 5 or wino3):
                # the u3 input is the same for every branch: its transform runs once, before the branch lanes fork
                # (5x5: F(wino_m, 5); the 'fast' mode's 3x3: F(4, 3))
                P.conv_winograd(p + "conva", cur_in, cat.chans(0, cmid), W(p + "conva.weight"), share_in=(uname == "u3"),
                                m=wino_stage.get(uname, wino_m) if k == 5 else wino3_m)
            else:
                P.conv(p + "conva", cur_in, cat.chans(0, cmid), W(p + "conva.weight"))
            c = cmid
            win = cat
            for i in range(units):
                q = p + "dense.units.%d." % i
                t1 = View(P.buf(q + "t1", win.h, win.w, arch.DENSE_MID))
                P.conv(q + "conv1", win.chans(0, c), t1, W(q + "conv1.weight"), bn=BN(q + "conv1/bn"), relu=1,
                       pre=BN(q + "preact_bna/bn"))
                win = win.crop(cm)          # crop_to_shape + cat (net_utils.py:147-148) == shrink the window
                P.conv(q + "conv2", t1, win.chans(c, arch.DENSE_GROWTH), W(q + "conv2.weight"), groups=arch.DENSE_GROUPS)
                c += arch.DENSE_GROWTH
            uo = View(P.buf(p + "out", win.h, win.w, ctot))
            P.conv(p + "convf", win.chans(0, ctot), uo, W(p + "convf.weight"), pre=BN(p + "dense.blk_bna.bn"))
            nxt = View(P.buf(p + "up", 2 * win.h, 2 * win.w, ctot))
            P.add(Op(OP_UPADD, p + "upadd", x=uo, res=skip, y=nxt))
            cur_in = nxt
        u1 = View(P.buf(pb + "u1", g["out"], g["out"], 64))
        if winograd and (k == 5 or wino3):
            P.conv_winograd(pb + "u1.conva", cur_in, u1, W(pb + "u1.conva.weight"), pad=_tf_same(cur_in.h, k, 1),
                            bn=BN(pb + "u0.bn"), relu=1, m=wino_stage.get("u1", wino_m) if k == 5 else wino3_m)
        else:
            P.conv(pb + "u1.conva", cur_in, u1, W(pb + "u1.conva.weight"), pad=_tf_same(cur_in.h, k, 1),
                   bn=BN(pb + "u0.bn"), relu=1)
        oc = arch.branch_out_ch(br, nr_types)
        lg = Buf("logits." + br, g["out"], g["out"], oc)
        P.logits[br] = lg
        heads[br] = P.add(Op(OP_HEAD, pb + "u0.conv", x=u1, y=View(lg),
                             w=np.ascontiguousarray(W(pb + "u0.conv.weight").reshape(oc, 64), np.float32),
                             bias=np.asarray(W(pb + "u0.conv.bias"), np.float32), cout=oc))
    if with_predmap:
        pm = Buf("pred_map", g["out"], g["out"], 3 if nr_types is None else 4)
        P.pred_map = pm
        P.add(Op(OP_PREDMAP, "infer_step.epilogue", y=View(pm), extra={"branches": list(arch.branch_names(nr_types))}))
    if x3 is None:
        x3 = 9 if conservative else int(os.environ.get("HVN_X3", "6"))       # default since round 4: six partial products (measured: the fp32-MFMA path's own error band)
    if x3:
        # (which pipe a layer's products run on is a rule by LAYER, never by what the chain pass did: HVN_CHAIN=0 runs the same bits unchained)
        P.mark_x3(int(x3), d1=os.environ.get("HVN_X3_D1", "1") != "0", chain=os.environ.get("HVN_X3_CHAIN", "d0"))
    if isinstance(chain, str) and chain.startswith("bf16"):
        # the bf16 path (csrc/hvn_conv_chain_bf16.hip): seams whose conv3 reduces over 64 or 128 channels in whole 64-channel slabs --
        # d0's three and d1's plain-residual ones; chain = "bf16:d0" | "bf16:d0d1" names the blocks whose seams are chained
        blocks = tuple(chain.split(":", 1)[1].replace("d0d1", "d0,d1").split(",")) if ":" in chain else ("d0", "d1")

        def bf16_seam(a, b):
            x2 = a.extra.get("x2")
            ka = a.x.c + (x2.c if x2 is not None else 0)
            return a.name.split(".")[0] in blocks and a.x.c % 64 == 0 and (x2 is None or x2.c % 64 == 0) and ka in (64, 128) and a.cout % 256 == 0
        P.fuse_chains(128, accept=bf16_seam)
    elif chain:
        P.fuse_chains(int(os.environ.get("HVN_CHAIN_MAXN2", "128")))
    if os.environ.get("HVN_FUSE_UPADD", "0") != "0":
        # measured (profiles/r03_upadd_fusion_ab.txt): parity-green and bit-equal, one launch and one tensor less per decoder stage, but
        # SLOWER -- the transform (64 register-resident tile values per thread) pays more for the second, quarter-rate address stream
        # than the up-sampled tensor's write + re-read cost: 61.15 vs 60.54 ms per step.  Off by default.
        P.fuse_upadd_into_winograd()
    # launch lanes: the decoder branches are independent between the shared u3 input and the
    # epilogue, so the engine may run them on concurrent streams (small dense-unit launches of one
    # branch then fill the chip together with the others')
    lanes, cur, start = [], None, 0
    for i, op in enumerate(P.ops):
        lane = op.name.split(".")[1] if (op.name.startswith("decoder.") and not op.extra.get("shared")) else "main"
        if lane != cur:
            if cur is not None:
                lanes.append((cur, start, i))
            cur, start = lane, i
    lanes.append((cur, start, len(P.ops)))
    # concurrent lanes must not share arena space: every buffer touched inside the branch region
    # stays live for the whole region
    br = [(lo, hi) for lane, lo, hi in lanes if lane != "main"]
    if br:
        r0, r1 = br[0][0], br[-1][1] - 1
        for b in P.bufs:
            if b.last >= r0 and b.first <= r1:
                b.first, b.last = min(b.first, r0), max(b.last, r1)
    P.pack()
    P.lanes = lanes
    return P

"""Lowering of the reference TRAINING step (/root/reference/models/hovernet/run_desc.py:12-109 over
net_desc.py:101-145 in train() mode) to a forward op list and the matching backward op list.

Unlike the inference plan (plan.py) nothing can be folded: BatchNorm runs on batch statistics, every conv output
and every BN-ReLU output is kept for the backward pass (288 GB of HBM: nothing is recomputed), and gradients
have their own buffers.  The rules that keep the op set small:

* Tensors are NHWC fp32, one buffer per tensor for the WHOLE batch (tensor-major arena, so a buffer's span stays
  below 4 GB at any batch size); crops, the dense-block concat and the skip crops are views (as in plan.py).
* Every gradient buffer is zeroed at the start of a step and every backward op ACCUMULATES into its
  destination, so fan-out (a tensor with several consumers) needs no add kernels.
* All running sums of a residual block (net_utils.py:263-264) share ONE gradient buffer: y = F(x) + x passes the
  gradient of y to x unchanged, so the `+ shortcut` backward is free.  The same aliasing gives the reference's
  `freeze` quirk for d0 (net_utils.py:250-266: only the units are scoped by no_grad, the shortcut conv and blk_bna
  still train) without a special case.
* The gradient buffer behind the output of a stride-2 conv is allocated DILATED ([2h, 2w]; the tensor's gradient is
  the step-2 view of it, the other entries stay zero): the data gradient of that conv is then a plain stride-1
  convolution of the dilated buffer with the flipped, transposed weights -- the same implicit-GEMM kernel as the
  forward pass (hvn_conv_igemm_f32), no transposed-conv kernel.
* The reference's grad scoping (net_desc.py:105-111) is followed literally: `grad` is on for conv0, d0's shortcut
  and blk_bna, conv_bot and the decoder always; for d0's units and d1..d3 only when `freeze` is False.

Consumers: engine/`train_engine.py` (HIP, the product) and tests/train_interp.py (torch-CPU interpreter, test only).
"""
from collections import OrderedDict

from . import arch


class GBuf:
    """Gradient storage [n, h, w, c] (alloc extent; may be shared by several tensors and may be dilated)."""

    def __init__(self, name, h, w, c):
        self.name, self.h, self.w, self.c = name, h, w, c
        self.off = None

    def size(self, n):
        return n * self.h * self.w * self.c


class TBuf:
    """Activation storage [n, h, w, c]; `g` = its gradient buffer (None: no gradient is ever needed)."""

    def __init__(self, name, h, w, c, g=None, gstep=1, dtype="f32"):
        self.name, self.h, self.w, self.c = name, h, w, c
        self.g, self.gstep, self.dtype = g, gstep, dtype
        self.off = None

    def size(self, n):
        return n * self.h * self.w * self.c


class TView:
    """Window of a buffer: element (y, x, ch) = buf[y0 + y*step, x0 + x*step, c0 + ch]."""

    def __init__(self, buf, y0=0, x0=0, h=None, w=None, c0=0, c=None, step=1):
        self.buf, self.y0, self.x0, self.c0, self.step = buf, y0, x0, c0, step
        self.h = buf.h if h is None else h
        self.w = buf.w if w is None else w
        self.c = buf.c if c is None else c

    def crop(self, m):
        return TView(self.buf, self.y0 + m * self.step, self.x0 + m * self.step, self.h - 2 * m, self.w - 2 * m, self.c0, self.c, self.step)

    def chans(self, c0, c):
        return TView(self.buf, self.y0, self.x0, self.h, self.w, self.c0 + c0, c, self.step)

    def grad(self):
        """The gradient view of a data view (None if its tensor carries no gradient)."""
        b = self.buf
        if b.g is None:
            return None
        s = b.gstep
        return TView(b.g, self.y0 * s, self.x0 * s, self.h, self.w, self.c0, self.c, self.step * s)

    def dense_cover(self):
        """Stride-1 view spanning the same rows/cols (for the dilated gradient of a stride-2 conv)."""
        return TView(self.buf, self.y0, self.x0, (self.h - 1) * self.step + 1, (self.w - 1) * self.step + 1, self.c0, self.c, 1)

    @property
    def req(self):
        return self.buf.g is not None


class TOp:
    def __init__(self, kind, name, **kw):
        self.kind, self.name = kind, name
        self.__dict__.update(kw)

    def __repr__(self):
        return "TOp(%s, %s)" % (self.kind, self.name)


def _tf_same(h, k, s):
    pad = max(k - s, 0) if h % s == 0 else max(k - (h % s), 0)
    return pad // 2, pad - pad // 2


class TrainPlan:
    """fwd / bwd op lists + buffers for one (mode, nr_types, freeze) configuration."""

    def __init__(self, mode="original", nr_types=None, freeze=False):
        assert mode in ("original", "fast")
        self.mode, self.nr_types, self.freeze = mode, nr_types, freeze
        self.geo = arch.geometry(mode)
        self.table = arch.param_table(mode, nr_types)
        self.bufs, self.gbufs, self.fwd, self.bwd = [], [], [], []
        self.convs = OrderedDict()      # weight key -> dict(cout, cin_g, kh, kw, groups, fwd_pack, dgrad_pack, train)
        self.bns = OrderedDict()        # bn key -> channels, in forward order (workspace slots)
        self.trainable = set()          # parameter keys that receive a gradient
        self.logits = OrderedDict()     # branch -> (channels, h, w): NCHW logits + their gradient, own small tensors
        self._grad_on = True
        self._build()
        self._backward()

    # -- tensors -------------------------------------------------------------------------------
    def new(self, name, h, w, c, req, gshare=None, dilate=False, dtype="f32"):
        g, gstep = None, 1
        if req:
            if gshare is not None:
                g, gstep = gshare
            else:
                gstep = 2 if dilate else 1
                g = GBuf(name + ".grad", h * gstep, w * gstep, c)
                self.gbufs.append(g)
        b = TBuf(name, h, w, c, g, gstep, dtype)
        self.bufs.append(b)
        return b

    # -- forward ops ---------------------------------------------------------------------------
    def conv(self, name, x, y, wkey, stride=1, pad=(0, 0), res=None, groups=1):
        cout, cin_g, kh, kw = self.table[wkey][1]
        assert cin_g * groups == x.c and cout == y.c, name
        assert y.h == (x.h + pad[0] + pad[1] - kh) // stride + 1, (name, x.h, y.h)
        train = self._grad_on
        info = self.convs.setdefault(wkey, dict(cout=cout, cin_g=cin_g, kh=kh, kw=kw, groups=groups, train=train, dgrad=False))
        if train:
            self.trainable.add(wkey)
        dx = train and x.req
        info["dgrad"] = info["dgrad"] or dx
        if res is not None and y.req:      # the residual add passes its gradient through a shared buffer
            gy, gr = y.grad(), res.grad()
            assert gr is not None and gr.buf is gy.buf and (gr.y0, gr.x0, gr.c0, gr.step) == (gy.y0, gy.x0, gy.c0, gy.step), name
        op = TOp("conv", name, x=x, y=y, res=res, wkey=wkey, kh=kh, kw=kw, stride=stride, pad=pad, groups=groups, train=train, dx=dx)
        self.fwd.append(op)
        return op

    def bnrelu(self, name, z, a, bnkey):
        assert (z.h, z.w, z.c) == (a.h, a.w, a.c) and self.table[bnkey + ".weight"][1] == (z.c,), name
        train = self._grad_on
        assert bnkey not in self.bns
        self.bns[bnkey] = z.c
        if train:
            self.trainable.update((bnkey + ".weight", bnkey + ".bias"))
        op = TOp("bnrelu", name, z=z, a=a, bnkey=bnkey, train=train)
        self.fwd.append(op)
        return op

    def req_out(self, *inputs, params=True):
        """Does the output of an op carry a gradient?  (grad mode on, and a trainable parameter or a grad input)"""
        return self._grad_on and (params or any(v.req for v in inputs))

    # -- the network (mirrors oracle/train_torch.forward_train) -----------------------------------
    def _res_block(self, name, cin, chans, units, stride, x, freeze_units):
        c1, c2, c3 = chans
        h_in = x.h
        h = h_in // stride
        outer = self._grad_on
        gs = None
        if outer:       # one gradient buffer for every running sum of the block (dilated when the block strides)
            g = GBuf(name + ".sum.grad", h * stride, h * stride, c3)
            self.gbufs.append(g)
            gs = (g, stride)
        s0 = TView(self.new(name + ".shortcut", h, h, c3, outer, gshare=gs))
        self.conv(name + ".shortcut", x, s0, name + ".shortcut.weight", stride=stride)
        prev = s0
        for i in range(units):
            p = "%s.units.%d." % (name, i)
            self._grad_on = outer and not freeze_units
            s = stride if i == 0 else 1
            f = x
            if i != 0:
                f = TView(self.new(p + "preact", h, h, c3, self.req_out()))
                self.bnrelu(p + "preact/bn", prev, f, p + "preact/bn")
            hi = f.h
            z1 = TView(self.new(p + "z1", hi, hi, c1, self.req_out()))
            self.conv(p + "conv1", f, z1, p + "conv1.weight")
            a1 = TView(self.new(p + "a1", hi, hi, c1, self.req_out()))
            self.bnrelu(p + "conv1/bn", z1, a1, p + "conv1/bn")
            z2 = TView(self.new(p + "z2", h, h, c2, self.req_out(), dilate=(s == 2)))
            self.conv(p + "conv2", a1, z2, p + "conv2.weight", stride=s, pad=_tf_same(hi, 3, s))
            a2 = TView(self.new(p + "a2", h, h, c2, self.req_out()))
            self.bnrelu(p + "conv2/bn", z2, a2, p + "conv2/bn")
            nxt = TView(self.new(p + "sum", h, h, c3, outer, gshare=gs))
            self.conv(p + "conv3", a2, nxt, p + "conv3.weight", res=prev)
            self._grad_on = outer
            prev = nxt
        out = TView(self.new(name + ".out", h, h, c3, self.req_out()))
        self.bnrelu(name + ".blk_bna", prev, out, name + ".blk_bna.bn")
        return out

    def _dense_block(self, name, cat, c_in, units, k):
        """cat: view over the concat buffer's first c_in channels at full extent (already written by conva)."""
        buf = cat.buf
        win = TView(buf, 0, 0, buf.h, buf.w, 0, c_in)
        c = c_in
        for i in range(units):
            p = "%s.units.%d." % (name, i)
            f = TView(self.new(p + "preact", win.h, win.w, c, True))
            self.bnrelu(p + "preact_bna", win, f, p + "preact_bna/bn")
            z1 = TView(self.new(p + "z1", win.h, win.w, arch.DENSE_MID, True))
            self.conv(p + "conv1", f, z1, p + "conv1.weight")
            a1 = TView(self.new(p + "a1", win.h, win.w, arch.DENSE_MID, True))
            self.bnrelu(p + "conv1/bn", z1, a1, p + "conv1/bn")
            nwin = win.crop((k - 1) // 2)
            self.conv(p + "conv2", a1, nwin.chans(c, arch.DENSE_GROWTH), p + "conv2.weight", groups=arch.DENSE_GROUPS)
            c += arch.DENSE_GROWTH
            win = TView(buf, nwin.y0, nwin.x0, nwin.h, nwin.w, 0, c)
        out = TView(self.new(name + ".out", win.h, win.w, c, True))
        self.bnrelu(name + ".blk_bna", win, out, name + ".blk_bna.bn")
        return out

    def _build(self):
        g, k = self.geo, self.geo["k"]
        frz = self.freeze
        self.img = TBuf("img", g["inp"], g["inp"], 3, dtype="u8")
        d0s = g["d"][0]
        z0 = TView(self.new("conv0.z", d0s, d0s, 64, True))
        self.conv0 = TOp("conv0", "conv0", x=TView(self.img), y=z0, wkey="conv0./.weight", pad=g["conv0_pad"], train=True)
        self.fwd.append(self.conv0)
        self.trainable.add("conv0./.weight")
        a0 = TView(self.new("conv0.a", d0s, d0s, 64, True))
        self.bnrelu("conv0.bn", z0, a0, "conv0.bn")
        d = []
        x = a0
        for name, cin, chans, units, stride in arch.RES_BLOCKS:
            if name == "d0":
                x = self._res_block(name, cin, chans, units, stride, x, freeze_units=frz)
            else:
                self._grad_on = not frz
                x = self._res_block(name, cin, chans, units, stride, x, freeze_units=False)
                self._grad_on = True
            d.append(x)
        d3 = TView(self.new("conv_bot.out", d[3].h, d[3].w, 1024, True))
        self.conv("conv_bot", d[3], d3, "conv_bot.weight")
        d[3] = d3
        d[0], d[1] = d[0].crop(g["crop0"]), d[1].crop(g["crop1"])
        for b in arch.branch_names(self.nr_types):
            p = "decoder.%s." % b
            lo, skips = d[3], (d[2], d[1])
            for uname, cin, cmid, units in (("u3", 1024, 256, 8), ("u2", 512, 128, 4)):
                skip = skips[0] if uname == "u3" else skips[1]
                q = p + uname + "."
                u = TView(self.new(q + "up", skip.h, skip.w, cin, True))
                self.fwd.append(TOp("upadd", q + "upadd", lo=lo, skip=skip, y=u))
                ctot = cmid + units * arch.DENSE_GROWTH
                hc = skip.h - (k - 1)
                gcat = GBuf(q + "cat.grad", hc, hc, ctot)
                self.gbufs.append(gcat)
                cat = self.new(q + "cat", hc, hc, ctot, True, gshare=(gcat, 1))
                self.conv(q + "conva", u, TView(cat).chans(0, cmid), q + "conva.weight")
                dense = self._dense_block(q + "dense", TView(cat).chans(0, cmid), cmid, units, k)
                lo = TView(self.new(q + "out", dense.h, dense.w, ctot, True))
                self.conv(q + "convf", dense, lo, q + "convf.weight")
            u = TView(self.new(p + "u1.up", d[0].h, d[0].w, 256, True))
            self.fwd.append(TOp("upadd", p + "u1.upadd", lo=lo, skip=d[0], y=u))
            z = TView(self.new(p + "u1.z", u.h, u.w, 64, True))
            self.conv(p + "u1.conva", u, z, p + "u1.conva.weight", pad=_tf_same(u.h, k, 1))
            a = TView(self.new(p + "u0.a", u.h, u.w, 64, True))
            self.bnrelu(p + "u0.bn", z, a, p + "u0.bn")
            co = arch.branch_out_ch(b, self.nr_types)
            self.logits[b] = (co, a.h, a.w)
            self.trainable.update((p + "u0.conv.weight", p + "u0.conv.bias"))
            self.fwd.append(TOp("head", p + "u0.conv", x=a, branch=b, wkey=p + "u0.conv.weight", bkey=p + "u0.conv.bias", cout=co))

    # -- backward list ---------------------------------------------------------------------------
    def _backward(self):
        for op in reversed(self.fwd):
            if op.kind == "head":
                self.bwd.append(TOp("head_bwd", op.name + ".bwd", x=op.x, dx=op.x.grad(), branch=op.branch, wkey=op.wkey, bkey=op.bkey, cout=op.cout))
            elif op.kind == "bnrelu":
                if op.train:
                    self.bwd.append(TOp("bnrelu_bwd", op.name + ".bwd", z=op.z, a=op.a, da=op.a.grad(), dz=op.z.grad(), bnkey=op.bnkey))
            elif op.kind == "conv":
                if not op.train:
                    continue
                gy = op.y.grad()
                self.bwd.append(TOp("wgrad", op.name + ".wgrad", x=op.x, dy=gy, wkey=op.wkey, kh=op.kh, kw=op.kw, stride=op.stride,
                                    pad=op.pad, groups=op.groups))
                if op.dx:
                    # data gradient = stride-1 conv of the (dilated) output gradient with flipped, transposed weights;
                    # a 1x1 stride-1 conv reads its output gradient through whatever step the view has
                    if op.kh == 1 and op.stride == 1:
                        dyv = gy
                    else:
                        assert gy.step == op.stride, op.name
                        dyv = gy.dense_cover()
                    self.bwd.append(TOp("dgrad", op.name + ".dgrad", dy=dyv, dx=op.x.grad(), wkey=op.wkey, kh=op.kh, kw=op.kw,
                                        pad=(op.kh - 1 - op.pad[0], op.kh - 1 - op.pad[1]), groups=op.groups))
            elif op.kind == "upadd":
                self.bwd.append(TOp("upadd_bwd", op.name + ".bwd", dy=op.y.grad(), dlo=op.lo.grad(), dskip=op.skip.grad()))
            elif op.kind == "conv0":
                self.bwd.append(TOp("conv0_wgrad", "conv0.wgrad", x=op.x, dy=op.y.grad(), wkey=op.wkey, pad=op.pad))
            else:  # pragma: no cover
                raise KeyError(op.kind)
        self._first_writers()

    def _first_writers(self):
        """Round 6 (round-5 verdict, missing #3): which backward ops are the FIRST writer of their destination in a step, and which
        gradient buffers therefore need no zero-fill.  Ops run in list order on one stream.  An op whose kernel has a store form
        (`bnrelu_bwd`: grad z; `dgrad`: grad x, the conv epilogue without its residual operand) gets `store = True` when no earlier op of
        the list wrote ANY element of the destination's buffer: storing v over the step's zero-fill equals adding it.  When such a first
        writer also covers the whole buffer (a BN's grad z and a conv's grad a are usually the only writer of their buffer), the buffer
        is never read before it is completely overwritten and `GBuf.zero` goes False: `layout` places those buffers behind
        `gzero_elems`, the part of the gradient arena the engine still clears each step.  tests/train_interp.py executes the same flags
        (stores, and NaN instead of zeros in the buffers that are not cleared) against the autograd oracle on the CPU."""
        touched = set()
        for g in self.gbufs:
            g.zero = True

        def first(v, has_store_form):
            if v is None:
                return False
            g = v.buf
            untouched = id(g) not in touched
            touched.add(id(g))
            if not (untouched and has_store_form):
                return False
            if (v.y0, v.x0, v.c0, v.step, v.h, v.w, v.c) == (0, 0, 0, 1, g.h, g.w, g.c):
                g.zero = False
            return True
        for op in self.bwd:
            if op.kind == "bnrelu_bwd":
                op.store = first(op.dz, True)
            elif op.kind == "dgrad":
                op.store = first(op.dx, True)
            elif op.kind == "head_bwd":
                first(op.dx, False)
            elif op.kind == "upadd_bwd":
                first(op.dlo, False)
                first(op.dskip, False)

    def wgrad_windows(self):
        """{index of a `wgrad` op in `bwd`: index of the first later op that WRITES the buffer its output gradient `dy` lives in, or None}.
        A weight gradient reads x (an activation: never written in the backward pass) and dy, and nothing in the backward pass reads what it
        writes -- so it may run on another stream from the moment dy is complete (the ops before it) to that later writer (the residual
        sums and the dense blocks' concats are accumulated further; a BN's grad z is not): `train_engine._floating_wgrads`."""
        out = {}
        for i, op in enumerate(self.bwd):
            if op.kind != "wgrad":
                continue
            buf, dead = op.dy.buf, None
            for j in range(i + 1, len(self.bwd)):
                if any(getattr(self.bwd[j], k, None) is not None and getattr(self.bwd[j], k).buf is buf for k in ("dx", "dz", "dlo", "dskip")):
                    dead = j
                    break
            out[i] = dead
        return out

    # -- memory ----------------------------------------------------------------------------------
    def layout(self, n):
        """Assign element offsets for batch `n` -> (data elements, grad elements); 64-element (256 B) aligned.  The gradient buffers
        that need the step's zero-fill come first (`gzero_elems` of them), the ones whose first writer overwrites them completely
        (`_first_writers`) after."""
        def place(items, off=0):
            for b in items:
                b.off = off
                off += (b.size(n) + 63) // 64 * 64
            return off
        self.gzero_elems = place([g for g in self.gbufs if g.zero])
        return place(self.bufs), place([g for g in self.gbufs if not g.zero], self.gzero_elems)

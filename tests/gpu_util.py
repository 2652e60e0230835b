"""Helpers for the -m gpu parity tests: tiny hand-made plans run through the real
libhvn_hip.so via hover_net_amd.engine.Engine, compared with tests/plan_interp.py."""
import numpy as np
import torch

from hover_net_amd import plan as PL


class MiniPlan(PL.Plan):
    def __init__(self):
        self.mode, self.nr_types = "mini", None
        self.geo = {"inp": 0, "out": 0}
        self.ops, self.bufs, self.logits, self.pred_map = [], [], {}, None
        self.image = PL.Buf("image", 1, 1, 3, "u8")
        self.arena_per_sample = 0


def rand_conv_weight(rng, cout, cin_g, k):
    return rng.normal(0.0, np.sqrt(2.0 / (cin_g * k * k)), (cout, cin_g, k, k))


def run_conv_case(n, xbuf_shape, xview, ybuf_shape, yview, wt, *, stride=1, pad=(0, 0), groups=1, bn=False, relu=0, pre=False,
                  res=False, post=False, seed=0, inplace_res=False, dtype="fp32", force_tile=None, x3=0):
    """Builds one CONV op over strided views, runs it on the GPU and with the torch
    interpreter.  xview / yview = (y0, x0, h, w, c0, c).  Returns (got, want) NHWC tensors of the
    WHOLE output buffer (so writes outside the view would be caught)."""
    from hover_net_amd.engine import Engine
    import plan_interp

    rng = np.random.default_rng(seed)
    P = MiniPlan()
    xb = P.buf("x", *xbuf_shape)
    yb = xb if ybuf_shape is None else P.buf("y", *ybuf_shape)
    xv = PL.View(xb, *xview)
    yv = PL.View(yb, *yview)
    cout = wt.shape[0]
    kw = {}
    if bn:
        kw["bn"] = (rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.2, cout))
    if pre:
        kw["pre"] = (rng.uniform(0.5, 1.5, xv.c), rng.normal(0, 0.3, xv.c))
    if post:
        kw["post"] = (rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.3, cout))
    rv = None
    if res:
        if inplace_res:
            rv = yv
        else:
            rb = P.buf("r", yb.h, yb.w, yb.c)
            rv = PL.View(rb, *yview)
        kw["res"] = rv
    op = P.conv("case", xv, yv, wt, stride=stride, pad=pad, groups=groups, relu=relu, **kw)
    if x3:
        assert op.tile_n in (128, 64) and groups == 1
        op.extra["x3"] = x3          # products on the bf16 matrix pipe from exact bf16x3 splits (csrc/hvn_conv_x3.hip)
    P.pack()
    eng = Engine(P, max_batch=n, dtype=dtype)
    g = torch.Generator().manual_seed(seed)
    if dtype == "bf16":
        # bf16 path: the arena holds bf16; the reference sees the same (rounded) inputs and bf16-rounded weights, so
        # what is left is the accumulation order and the rounding of the output to bf16
        eng.arena.view(torch.bfloat16).copy_(torch.randn(eng.arena.shape, generator=g))
        op.w = np.ascontiguousarray(torch.from_numpy(op.w).to(torch.bfloat16).float().numpy())
        A = plan_interp.Arena(P, n)
        A.flat.copy_(eng.arena.view(torch.bfloat16).float().cpu())
    else:
        eng.arena.copy_(torch.randn(eng.arena.shape, generator=g))
        A = plan_interp.Arena(P, n)
        A.flat.copy_(eng.arena.cpu())
    if force_tile is not None:
        eng.ops[0].tile_n = force_tile
    eng.run_raw(n)
    torch.cuda.synchronize()
    r = A.view(op.res).clone() if op.res is not None else None
    A.view(op.y).copy_(plan_interp.conv_ref(op, A.view(op.x).clone(), r))
    b = yb
    arena = eng.arena.view(torch.bfloat16).float().cpu() if dtype == "bf16" else eng.arena.cpu()
    got = arena[:, b.offset:b.offset + b.size].view(n, b.h, b.w, b.c)
    want = A.tensor(b)
    return got, want

"""GPU: OP_CHAIN (csrc/hvn_conv_chain.hip) -- a residual unit's conv3 (+ residual / fused shortcut / block-closing BN-ReLU) chained with
the next unit's pre-activation + conv1 (/root/reference/models/hovernet/net_utils.py:250-266) -- against the torch interpreter of the
op's own fields, and BIT-EQUAL to the two CONV launches it replaces (same reduction order per output element)."""
import numpy as np
import pytest
import torch

import plan_interp
from gpu_util import MiniPlan, rand_conv_weight
from hover_net_amd import plan as PL

pytestmark = pytest.mark.gpu


def _two_convs(n, h, w, k1, c, n2, *, res, x2, post, pre, seed, inplace=True):
    """conv3-like (k1 [+ x2] -> c, + res, [post]) followed by conv1-like ([pre] c -> n2, bn, relu), over a (h, w) grid."""
    rng = np.random.default_rng(seed)
    P = MiniPlan()
    t2 = PL.View(P.buf("t2", h, w, k1))
    acc = PL.View(P.buf("acc", h, w, c))
    out = acc if inplace else PL.View(P.buf("out", h, w, c))
    t1 = PL.View(P.buf("t1", h, w, n2))
    kw = {}
    if x2:
        k1b, s2 = x2
        xin = PL.View(P.buf("block_in", (h - 1) * s2 + 1, (w - 1) * s2 + 2, k1b))      # one surplus column: the view is wider than the grid
        kw.update(x2=xin, wt2=rand_conv_weight(rng, c, k1b, 1), stride2=s2)
    if res:
        kw["res"] = acc
    if post:
        kw["post"] = (rng.uniform(0.5, 1.5, c), rng.normal(0, 0.3, c))
    P.conv("u.conv3", t2, out, rand_conv_weight(rng, c, k1, 1), **kw)
    P.conv("v.conv1", out, t1, rand_conv_weight(rng, n2, c, 1), bn=(rng.uniform(0.5, 1.5, n2), rng.normal(0, 0.2, n2)), relu=1,
           pre=(rng.uniform(0.5, 1.5, c), rng.normal(0, 0.3, c)) if pre else None)
    return P, (t2, acc, out, t1)


def _run(P, n, seed, bm=None):
    """Every buffer filled with values that depend on (seed, buffer name) only -- the fused and the unfused plan pack their arenas
    differently, but see the same tensors."""
    from hover_net_amd.engine import Engine

    P.pack()
    eng = Engine(P, max_batch=n)
    eng.arena.fill_(float("nan"))
    names = sorted(b.name for b in P.bufs)
    for b in sorted(P.bufs, key=lambda b: -b.first):      # later-born buffers first: a pure output may share memory with a dead input
        g = torch.Generator().manual_seed(seed * 1000 + names.index(b.name))
        eng.buffer(PL.View(b), n).copy_(torch.randn((n, b.h, b.w, b.c), generator=g))
    start = eng.arena.cpu().clone()
    if bm is not None:                                # CHAIN: pixels per workgroup (the engine's timing pass would pick)
        eng.ops[0].tile_n = bm
    eng.run_raw(n)
    torch.cuda.synchronize()
    return eng, start


CASES = [
    # n, h, w, k1, c, n2, res, x2 (k1b, stride2), post, pre
    (2, 24, 24, 64, 256, 64, True, None, False, True),        # d0 units 1..2 -> next conv1 (exact multiple of the 128-pixel tile)
    (3, 19, 23, 64, 256, 64, True, None, False, True),        # ragged: tiles straddle samples, tail rows past M
    (2, 17, 21, 64, 256, 64, False, (64, 1), False, True),    # d0 unit 0: fused shortcut, no residual
    (2, 15, 18, 64, 256, 128, True, None, True, False),       # d0 last unit: block-closing BN-ReLU, feeds d1 unit 0 (no pre-activation)
    (2, 13, 11, 128, 512, 128, False, (256, 2), False, True), # d1 unit 0: strided shortcut appended to k (K = 384, 12 k-steps)
    (1, 16, 16, 128, 512, 128, True, None, False, True),      # d1 units 1..3
    (1, 9, 10, 32, 64, 64, False, (32, 1), False, True),      # smallest legal shape: K = 32 + 32, one 64-channel chunk
]


@pytest.mark.parametrize("bm", [128, 64])
@pytest.mark.parametrize("case", CASES)
def test_chain_matches_reference_and_unfused_launches(case, bm):
    n, h, w, k1, c, n2, res, x2, post, pre = case
    # fused
    P, (t2, acc, out, t1) = _two_convs(n, h, w, k1, c, n2, res=res, x2=x2, post=post, pre=pre, seed=7, inplace=not post)
    P.fuse_chains()
    assert [o.kind for o in P.ops] == [PL.OP_CHAIN]
    eng, start = _run(P, n, seed=11, bm=bm)
    got = eng.arena.cpu()
    # torch interpreter on the same arena contents, from the CHAIN op's own fields
    A = plan_interp.Arena(P, n)
    A.flat.copy_(start)
    op = P.ops[0]
    r = A.view(op.res).clone() if op.res is not None else None
    xx2 = A.view(op.extra["x2"]).clone() if op.extra.get("x2") is not None else None
    y, y2 = plan_interp.chain_ref(op, A.view(op.x).clone(), r, xx2)
    A.view(op.y).copy_(y)
    A.view(op.extra["y2"]).copy_(y2)
    want = A.flat
    live = ~torch.isnan(want)
    assert torch.equal(live, ~torch.isnan(got))
    scale = float(want[live].abs().max())
    assert float((got[live] - want[live]).abs().max()) < 2e-5 * max(1.0, scale)
    # nothing outside the two output views was touched
    untouched = torch.ones_like(start, dtype=torch.bool)
    for v in (op.y, op.extra["y2"]):
        b = v.buf
        m = untouched[:, b.offset:b.offset + b.size].view(n, b.h, b.w, b.c)
        m[:, v.y0:v.y0 + v.h, v.x0:v.x0 + v.w, v.c0:v.c0 + v.c] = False
    untouched &= live
    assert torch.equal(got[untouched], start[untouched])
    # the two separate launches: same bits in both outputs
    Q, (_, _, out_q, t1_q) = _two_convs(n, h, w, k1, c, n2, res=res, x2=x2, post=post, pre=pre, seed=7, inplace=not post)
    assert [o.kind for o in Q.ops] == [PL.OP_CONV, PL.OP_CONV]
    eng2, _ = _run(Q, n, seed=11)
    assert torch.equal(eng2.buffer(out_q, n), eng.buffer(out, n))
    assert torch.equal(eng2.buffer(t1_q, n), eng.buffer(t1, n))


X3R_CASES = [  # + shapes only csrc/hvn_conv_chain_x3r.hip's walk distinguishes (32-channel chunks: cout = 64 is two of them)
    (2, 9, 10, 64, 64, 64, True, None, False, True),          # two chunks, ragged tail
    (1, 31, 33, 64, 128, 128, True, None, True, True),        # block-closing BN-ReLU AND a pre-activation, cout2 = 128
    (2, 12, 14, 64, 256, 64, False, (64, 2), False, True),    # fused shortcut sampled at stride 2
    (1, 7, 5, 64, 64, 64, True, None, False, False),          # one partial tile; neither BN-ReLU
]


def _x3r_exists(case):
    n, h, w, k1, c, n2, res, x2, post, pre = case
    return k1 == 64 and (x2 is None or (x2[0] == 64 and not res and n2 == 64))


@pytest.mark.parametrize("terms", [6, 9])
@pytest.mark.parametrize("form", ["x3", "x3r"])
@pytest.mark.parametrize("case", CASES + X3R_CASES)
def test_chain_on_the_bf16_pipe_matches_reference_and_unfused_bf16x3_launches(case, form, terms):
    """csrc/hvn_conv_chain_x3.hip: the same op with both GEMMs' products on the bf16 matrix pipe (bf16x3 splits).  Against the torch
    interpreter at the bf16x3 kernel's tolerance, and BIT-EQUAL to the two bf16x3 CONV launches it replaces.  form "x3r": the same
    through csrc/hvn_conv_chain_x3r.hip (input tile resident in registers, operands a chunk ahead in flight; hvn_op.tile_n = X3R) where
    that form exists -- and a refused launch where it does not."""
    from hover_net_amd import lib as L
    from hover_net_amd.engine import X3R

    n, h, w, k1, c, n2, res, x2, post, pre = case
    if form == "x3" and case in X3R_CASES and terms == 9:
        pytest.skip("covered with six terms")

    def build(fuse):
        P, views = _two_convs(n, h, w, k1, c, n2, res=res, x2=x2, post=post, pre=pre, seed=7, inplace=not post)
        for o in P.ops:
            o.extra["x3"] = terms
            o.extra["x3_chain"] = True
        if fuse:
            P.fuse_chains()
        return P, views

    P, (t2, acc, out, t1) = build(True)
    assert [o.kind for o in P.ops] == [PL.OP_CHAIN] and P.ops[0].extra["x3"] == terms
    if form == "x3r" and not _x3r_exists(case):
        with pytest.raises(L.HvnError):
            _run(P, n, seed=11, bm=X3R)
        return
    eng, start = _run(P, n, seed=11, bm=X3R if form == "x3r" else 128)
    assert eng.ops[0].act_dtype == (2 if terms == 9 else 3) and eng.ops[0].tile_n == (X3R if form == "x3r" else 128)
    got = eng.arena.cpu()
    A = plan_interp.Arena(P, n)
    A.flat.copy_(start)
    op = P.ops[0]
    r = A.view(op.res).clone() if op.res is not None else None
    xx2 = A.view(op.extra["x2"]).clone() if op.extra.get("x2") is not None else None
    y, y2 = plan_interp.chain_ref(op, A.view(op.x).clone(), r, xx2)
    A.view(op.y).copy_(y)
    A.view(op.extra["y2"]).copy_(y2)
    want = A.flat
    live = ~torch.isnan(want)
    assert torch.equal(live, ~torch.isnan(got))
    scale = float(want[live].abs().max())
    assert float((got[live] - want[live]).abs().max()) < 6e-5 * max(1.0, scale)
    untouched = torch.ones_like(start, dtype=torch.bool)
    for v in (op.y, op.extra["y2"]):
        b = v.buf
        m = untouched[:, b.offset:b.offset + b.size].view(n, b.h, b.w, b.c)
        m[:, v.y0:v.y0 + v.h, v.x0:v.x0 + v.w, v.c0:v.c0 + v.c] = False
    untouched &= live
    assert torch.equal(got[untouched], start[untouched])
    Q, (_, _, out_q, t1_q) = build(False)
    assert [o.kind for o in Q.ops] == [PL.OP_CONV, PL.OP_CONV]
    eng2, _ = _run(Q, n, seed=11)
    assert all(o.act_dtype in (2, 3) for o in eng2.ops)
    assert torch.equal(eng2.buffer(out_q, n), eng.buffer(out, n))
    assert torch.equal(eng2.buffer(t1_q, n), eng.buffer(t1, n))


def test_network_with_and_without_chains_is_bit_equal(monkeypatch):
    from hover_net_amd import net_desc
    from hover_net_amd.synth import synth_state_dict, synth_tiles

    tiles = torch.from_numpy(synth_tiles(2, 270, seed=3)).cuda()
    from hover_net_amd.engine import X3R

    outs = []
    for chain, x3r in (("1", "force"), ("1", "0"), ("0", "0")):     # chained seams on hvn_conv_chain_x3r.hip | hvn_conv_chain_x3.hip | two launches
        monkeypatch.setenv("HVN_CHAIN", chain)
        monkeypatch.setenv("HVN_CHAIN_X3R", x3r)
        net = net_desc.create_model(mode="original", nr_types=5, input_ch=3)
        net.load_state_dict(synth_state_dict("original", 5, seed=2), strict=True)
        net = net.cuda().eval()
        eng = net.engine(2)
        assert any(o.kind == PL.OP_CHAIN for o in eng.plan.ops) == (chain == "1")
        on_x3r = [i for i, o in enumerate(eng.plan.ops) if o.kind == PL.OP_CHAIN and eng.ops[i].tile_n == X3R]
        assert (len(on_x3r) == 3) == (x3r == "force"), on_x3r          # d0's three seams all have the form
        logits, pred = eng.run(tiles)
        outs.append({k: v.clone() for k, v in logits.items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[0][k], outs[2][k]), k


def test_network_with_and_without_upadd_fusion_is_bit_equal(monkeypatch):
    """UPADD fused into the Winograd input transform (Plan.fuse_upadd_into_winograd, HVN_FUSE_UPADD=1 -- measured slower, so not the
    default: nearest2x(lo) + skip formed on the fly) gives the bits of the separate UPADD launch followed by WINO_IN, in both decoder
    geometries."""
    from hover_net_amd import net_desc
    from hover_net_amd.synth import synth_state_dict, synth_tiles

    for mode, nt, size in (("original", 5, 270), ("fast", None, 256)):
        tiles = torch.from_numpy(synth_tiles(2, size, seed=4)).cuda()
        outs = []
        for fuse in ("1", "0"):
            monkeypatch.setenv("HVN_FUSE_UPADD", fuse)
            net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3)
            net.load_state_dict(synth_state_dict(mode, nt, seed=2), strict=True)
            net = net.cuda().eval()
            eng = net.engine(2)
            assert any(o.kind == PL.OP_UPADD for o in eng.plan.ops) == (fuse == "0")
            logits, _ = eng.run(tiles)
            outs.append({k: v.clone() for k, v in logits.items()})
        for k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), (mode, k)


def test_network_on_sub_batch_and_branch_streams_is_bit_equal(monkeypatch):
    """HVN_SPLIT=2 HVN_LANES=2 -- the encoder as two sub-batches on two streams, the decoder branches on their own streams
    (engine.Engine.run; +2.6 % on the round-3 kernels, profiles/r03_streams_ab.txt) -- is a launch SCHEDULE: every kernel sees the
    same per-sample operands, so logits and prediction map carry the bits of the single-stream run (chains and Winograd scratch
    included: each sample owns its rows of every buffer)."""
    from hover_net_amd import net_desc
    from hover_net_amd.synth import synth_state_dict, synth_tiles

    tiles = torch.from_numpy(synth_tiles(5, 270, seed=6)).cuda()       # 5: uneven sub-batches (2 + 3)
    outs = []
    for split, lanes in (("1", "0"), ("2", "2")):
        monkeypatch.setenv("HVN_SPLIT", split)
        monkeypatch.setenv("HVN_LANES", lanes)
        net = net_desc.create_model(mode="original", nr_types=5, input_ch=3)
        net.load_state_dict(synth_state_dict("original", 5, seed=2), strict=True)
        net = net.cuda().eval()
        eng = net.engine(5)
        assert eng.n_split == int(split) and eng.n_lane_streams == int(lanes)
        logits, pred = eng.run(tiles)
        torch.cuda.synchronize()
        out = {k: v.clone() for k, v in logits.items()}
        out["pred_map"] = pred.clone()
        outs.append(out)
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k

"""-m gpu: the fused implicit-GEMM conv kernel (hover_net_amd/csrc/hvn_conv.hip) through the
C ABI, against the torch-CPU interpretation of the same descriptor, on every conv class of
SURVEY.md 2.2(i) at small spatial sizes.  fp32 MFMA is an exact fmaf chain, so the only
difference to torch is summation order: tolerance 2e-4 abs on O(1) outputs (K up to 25.6k)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-4


def _check(got, want, tol=TOL):
    assert got.shape == want.shape
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err <= tol, "max abs err %g" % err


def _case(**kw):
    from gpu_util import run_conv_case

    return run_conv_case(**kw)


def _w(cout, cin_g, k, seed=1):
    from gpu_util import rand_conv_weight

    return rand_conv_weight(np.random.default_rng(seed), cout, cin_g, k)


@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 256), (256, 64), (128, 512), (2048, 1024), (32, 32), (288, 128)])
def test_conv1x1(cin, cout):
    n, s = 2, 13  # M = 338: exercises the M tail (not a multiple of 128)
    got, want = _case(n=n, xbuf_shape=(s, s, cin), xview=(0, 0, s, s, 0, cin), ybuf_shape=(s, s, cout),
                      yview=(0, 0, s, s, 0, cout), wt=_w(cout, cin, 1), bn=True, relu=1)
    _check(got, want)


def test_conv1x1_prologue_residual_post():
    # residual-unit conv3 with the block-closing BN-ReLU, conv1 with a pre-activation prologue
    n, s = 2, 17
    got, want = _case(n=n, xbuf_shape=(s, s, 64), xview=(0, 0, s, s, 0, 64), ybuf_shape=(s, s, 256),
                      yview=(0, 0, s, s, 0, 256), wt=_w(256, 64, 1), res=True, post=True)
    _check(got, want)
    got, want = _case(n=n, xbuf_shape=(s, s, 256), xview=(0, 0, s, s, 0, 256), ybuf_shape=(s, s, 64),
                      yview=(0, 0, s, s, 0, 64), wt=_w(64, 256, 1), pre=True, bn=True, relu=1)
    _check(got, want)


def test_conv1x1_inplace_residual():
    n, s = 1, 20
    got, want = _case(n=n, xbuf_shape=(s, s, 64), xview=(0, 0, s, s, 0, 64), ybuf_shape=(s, s, 128),
                      yview=(0, 0, s, s, 0, 128), wt=_w(128, 64, 1), res=True, inplace_res=True)
    _check(got, want)


def test_conv1x1_stride2_shortcut():
    n, s = 2, 24
    got, want = _case(n=n, xbuf_shape=(s, s, 256), xview=(0, 0, s, s, 0, 256), ybuf_shape=(12, 12, 512),
                      yview=(0, 0, 12, 12, 0, 512), wt=_w(512, 256, 1), stride=2)
    _check(got, want)


@pytest.mark.parametrize("stride,pad,s,so", [(1, (1, 1), 18, 18), (2, (0, 1), 18, 9)])
def test_conv3x3_tf_same(stride, pad, s, so):
    n = 2
    got, want = _case(n=n, xbuf_shape=(s, s, 128), xview=(0, 0, s, s, 0, 128), ybuf_shape=(so, so, 128),
                      yview=(0, 0, so, so, 0, 128), wt=_w(128, 128, 3), stride=stride, pad=pad, bn=True, relu=1)
    _check(got, want)


def test_conv5x5_valid_big_k():
    # u3.conva class: 1024 -> 256, K = 25600
    n, s = 1, 12
    got, want = _case(n=n, xbuf_shape=(s, s, 1024), xview=(0, 0, s, s, 0, 1024), ybuf_shape=(8, 8, 512),
                      yview=(0, 0, 8, 8, 0, 256), wt=_w(256, 1024, 5))
    _check(got, want)


def test_conv5x5_same_pad():
    n, s = 1, 14
    got, want = _case(n=n, xbuf_shape=(s, s, 256), xview=(0, 0, s, s, 0, 256), ybuf_shape=(s, s, 64),
                      yview=(0, 0, s, s, 0, 64), wt=_w(64, 256, 5), pad=(2, 2), bn=True, relu=1)
    _check(got, want)


@pytest.mark.parametrize("k", [5, 3])
def test_dense_unit_views(k):
    # dense-block conv1 reads a cropped window of the concat buffer (first 288 of 512 channels),
    # the grouped conv2 (block-diagonal packed) writes 32 channels at offset 288 of a smaller window
    n, s = 2, 16
    c = (k - 1) // 2
    got, want = _case(n=n, xbuf_shape=(s, s, 512), xview=(2, 2, s - 4, s - 4, 0, 288), ybuf_shape=(s - 4, s - 4, 128),
                      yview=(0, 0, s - 4, s - 4, 0, 128), wt=_w(128, 288, 1), pre=True, bn=True, relu=1)
    _check(got, want)
    so = s - 4 - (k - 1)
    got, want = _case(n=n, xbuf_shape=(s - 4, s - 4, 128), xview=(0, 0, s - 4, s - 4, 0, 128), ybuf_shape=(s, s, 512),
                      yview=(2 + c, 2 + c, so, so, 288, 32), wt=_w(32, 32, k), groups=4)
    _check(got, want)
    # nothing outside the 32-channel window may have been touched
    assert torch.equal(got[..., :288], want[..., :288]) and torch.equal(got[..., 320:], want[..., 320:])


def test_descriptor_validation():
    import ctypes

    from hover_net_amd import lib as L

    op = L.hvn_op()
    op.kind = 2
    assert L.lib().hvn_run_op(ctypes.addressof(op), 1, None) == -1
    assert b"conv" in L.lib().hvn_last_error()
    op.kind = 99
    assert L.lib().hvn_run_op(ctypes.addressof(op), 1, None) == -1


@pytest.mark.parametrize("stride2,cin2,cout", [(1, 64, 256), (2, 256, 512)])
def test_conv1x1_fused_shortcut_second_input(stride2, cin2, cout):
    """Residual block unit 0: conv3(t2) + strided 1x1 shortcut(x_in) as ONE GEMM over two inputs, with the
    block-closing BN-ReLU epilogue."""
    import plan_interp
    from gpu_util import MiniPlan, rand_conv_weight
    from hover_net_amd import plan as PL
    from hover_net_amd.engine import Engine

    rng = np.random.default_rng(5)
    n, so = 2, 13
    si = (so - 1) * stride2 + 1 + (1 if stride2 == 2 else 0)   # even input like the real net (H = 2*Ho)
    P = MiniPlan()
    x = PL.View(P.buf("t2", so, so, 64))
    x2 = PL.View(P.buf("xin", si, si, cin2))
    y = PL.View(P.buf("y", so, so, cout))
    op = P.conv("fused", x, y, rand_conv_weight(rng, cout, 64, 1), x2=x2, wt2=rand_conv_weight(rng, cout, cin2, 1),
                stride2=stride2, post=(rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.3, cout)))
    P.pack()
    eng = Engine(P, max_batch=n, n_split=1)
    eng.arena.copy_(torch.randn(eng.arena.shape, generator=torch.Generator().manual_seed(1)))
    A = plan_interp.Arena(P, n)
    A.flat.copy_(eng.arena.cpu())
    eng.run_raw(n)
    torch.cuda.synchronize()
    want = plan_interp.conv_ref(op, A.view(op.x).clone(), None, A.view(x2).clone())
    got = eng.buffer(op.y, n).cpu()
    _check(got, want)


@pytest.mark.parametrize("m,k", [(2, 5), (4, 5), (4, 3), (6, 3), (6, 5)])
@pytest.mark.parametrize("cin,cout,s,pad,bn", [(64, 128, 14, (0, 0), False), (256, 64, 12, (2, 2), True), (1024, 256, 10, (0, 0), False),
                                               (64, 32, 17, (0, 0), True), (512, 512, 33, (1, 1), True)])
def test_winograd_5x5_matches_direct(cin, cout, s, pad, bn, m, k):
    """WINO_IN -> batched GEMM -> WINO_OUT (F(m x m, k x k): F(2,5), F(4,5), F(4,3), F(6,3), F(6,5)) against a direct fp32
    convolution, writing into a channel window of a wider buffer like the dense-block concat; output extents that
    are not a multiple of m exercise the partial last tile."""
    import plan_interp
    import torch.nn.functional as F
    from gpu_util import MiniPlan, rand_conv_weight
    from hover_net_amd import plan as PL
    from hover_net_amd.engine import Engine

    rng = np.random.default_rng(7)
    n = 2
    so = s + pad[0] + pad[1] - (k - 1)
    P = MiniPlan()
    x = PL.View(P.buf("x", s, s, cin))
    ybuf = P.buf("y", so, so, cout + 32)
    y = PL.View(ybuf, 0, 0, so, so, 32, cout)
    wt = rand_conv_weight(rng, cout, cin, k)
    kw = dict(bn=(rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.2, cout)), relu=1) if bn else {}
    P.conv_winograd("w", x, y, wt, pad=pad, m=m, **kw)
    ybuf.first = 0      # keep the output buffer live from the start so the packer cannot lend its space to V / M
    P.pack()
    eng = Engine(P, max_batch=n, n_split=1)
    eng.arena.copy_(torch.randn(eng.arena.shape, generator=torch.Generator().manual_seed(3)))
    xin = eng.buffer(x, n).cpu().clone()
    before = eng.buffer(PL.View(ybuf), n).cpu().clone()
    eng.run_raw(n)
    torch.cuda.synchronize()
    w = torch.from_numpy(wt).float()
    want = F.conv2d(F.pad(xin.permute(0, 3, 1, 2), (pad[0], pad[1], pad[0], pad[1])), w).permute(0, 2, 3, 1)
    if bn:
        want = F.relu(want * torch.from_numpy(kw["bn"][0]).float() + torch.from_numpy(kw["bn"][1]).float())
    got = eng.buffer(PL.View(ybuf), n).cpu()
    # F(6x6, 5x5) (optional, HVN_WINOGRAD=6): ten interpolation points, ~15x the fp32 error of F(4x4, 5x5) (hover_net_amd/winograd.py)
    _check(got[..., 32:], want, tol=4e-3 if (m, k) == (6, 5) else 5e-4)
    assert torch.equal(got[..., :32], before[..., :32])      # the neighbouring channels are untouched
    # and the interpreter's transform-domain tensors agree stage by stage
    A = plan_interp.Arena(P, n)
    A.view(x).copy_(xin)
    v = plan_interp.wino_in_ref(P.ops[0], A.view(x).clone())
    _check(eng.buffer(P.ops[0].y, n).cpu().reshape(v.shape), v, tol=2e-3 if (m, k) == (6, 5) else 1e-4)


@pytest.mark.parametrize("k,pad,pre", [(3, (1, 1), False), (1, (0, 0), True), (5, (2, 2), False)])
def test_256x64_tile_equals_128x64_tile(k, pad, pre):
    """The 256 x 64 workgroup tile of the 64-channel layers (tile_n = 64 | 0x100, picked per shape by Engine.autotune_tiles) walks k
    in the same order per output element as the 128 x 64 tile: same bits, and the torch interpreter's values."""
    from gpu_util import rand_conv_weight, run_conv_case
    n, s = 3, 21                                  # 3 * 21 * 21 = 1323 pixels: 5.2 tiles of 256, ragged, straddling samples
    wt = rand_conv_weight(np.random.default_rng(3), 64, 64, k)
    kw = dict(n=n, xbuf_shape=(s, s, 64), xview=(0, 0, s, s, 0, 64), ybuf_shape=(s, s, 64), yview=(0, 0, s, s, 0, 64), wt=wt, pad=pad,
              bn=True, relu=1, pre=pre, res=not pre, seed=5)
    got64, want = run_conv_case(force_tile=64, **kw)
    got256, _ = run_conv_case(force_tile=320, **kw)
    assert torch.equal(got64, got256)
    assert float((got256 - want).abs().max()) < 2e-4 * max(1.0, float(want.abs().max()))

"""-m gpu: csrc/hvn_conv_x3.hip -- the fp32 implicit-GEMM convolution whose PRODUCTS run on the bf16 matrix pipe from exact three-way
bf16 splits of the fp32 operands (an fp32 value = h + m + l with 8 significand bits each; a bf16 x bf16 product is exact in fp32).
With all nine partial products per product the result is the fp32 dot product in another summation order, so the kernel is held to
the SAME tolerance as the fp32-MFMA kernel (tests/test_gpu_conv.py: 2e-4 abs on O(1) outputs at K up to 25.6k); the six-term form
drops the three partial products below 2^-24 of a product and gets 1.5x that.  Same classes as test_gpu_conv.py: 1x1 / 3x3 / 5x5,
strides, TF-same padding, prologue, residual (also in place), block BN-ReLU, the fused strided shortcut, the batched Winograd-domain
product, ragged pixel and channel tails, both column tiles (same bits).  Then the whole network against the reference-made goldens
(1e-3 on the logits, BASELINE north_star) with the plan's MFMA-bound launches on this kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {9: 2e-4, 6: 3e-4}


def _check(got, want, tol):
    assert got.shape == want.shape and torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err <= tol, "max abs err %g" % err


def _w(cout, cin_g, k, seed=1):
    from gpu_util import rand_conv_weight

    return rand_conv_weight(np.random.default_rng(seed), cout, cin_g, k)


def _case(**kw):
    from gpu_util import run_conv_case

    return run_conv_case(**kw)


@pytest.mark.parametrize("terms", [9, 6])
@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 256), (256, 64), (128, 512), (2048, 1024), (288, 128)])
def test_conv1x1(cin, cout, terms):
    n, s = 2, 13  # M = 338: exercises the pixel tail
    got, want = _case(n=n, xbuf_shape=(s, s, cin), xview=(0, 0, s, s, 0, cin), ybuf_shape=(s, s, cout), yview=(0, 0, s, s, 0, cout),
                      wt=_w(cout, cin, 1), bn=True, relu=1, x3=terms)
    _check(got, want, TOL[terms])


@pytest.mark.parametrize("terms", [9, 6])
def test_prologue_residual_post_inplace_stride(terms):
    n, s = 2, 17
    got, want = _case(n=n, xbuf_shape=(s, s, 64), xview=(0, 0, s, s, 0, 64), ybuf_shape=(s, s, 256), yview=(0, 0, s, s, 0, 256),
                      wt=_w(256, 64, 1), res=True, post=True, x3=terms)
    _check(got, want, TOL[terms])
    got, want = _case(n=n, xbuf_shape=(s, s, 256), xview=(0, 0, s, s, 0, 256), ybuf_shape=(s, s, 64), yview=(0, 0, s, s, 0, 64),
                      wt=_w(64, 256, 1), pre=True, bn=True, relu=1, x3=terms)
    _check(got, want, TOL[terms])
    got, want = _case(n=1, xbuf_shape=(20, 20, 64), xview=(0, 0, 20, 20, 0, 64), ybuf_shape=(20, 20, 128), yview=(0, 0, 20, 20, 0, 128),
                      wt=_w(128, 64, 1), res=True, inplace_res=True, x3=terms)
    _check(got, want, TOL[terms])
    got, want = _case(n=2, xbuf_shape=(24, 24, 256), xview=(0, 0, 24, 24, 0, 256), ybuf_shape=(12, 12, 512), yview=(0, 0, 12, 12, 0, 512),
                      wt=_w(512, 256, 1), stride=2, x3=terms)
    _check(got, want, TOL[terms])


@pytest.mark.parametrize("terms", [9, 6])
@pytest.mark.parametrize("stride,pad,s,so", [(1, (1, 1), 18, 18), (2, (0, 1), 18, 9)])
def test_conv3x3_tf_same(stride, pad, s, so, terms):
    got, want = _case(n=2, xbuf_shape=(s, s, 128), xview=(0, 0, s, s, 0, 128), ybuf_shape=(so, so, 128), yview=(0, 0, so, so, 0, 128),
                      wt=_w(128, 128, 3), stride=stride, pad=pad, bn=True, relu=1, x3=terms)
    _check(got, want, TOL[terms])


@pytest.mark.parametrize("terms", [9, 6])
def test_conv5x5_valid_big_k_into_a_channel_window(terms):
    # u3.conva class as a direct convolution: 1024 -> 256, K = 25600, written into a channel window of a wider (concat) buffer
    got, want = _case(n=1, xbuf_shape=(12, 12, 1024), xview=(0, 0, 12, 12, 0, 1024), ybuf_shape=(8, 8, 512), yview=(0, 0, 8, 8, 0, 256),
                      wt=_w(256, 1024, 5), x3=terms)
    _check(got, want, TOL[terms])
    # cropped input window + 64-wide plan tile (d0's 3x3 class)
    got, want = _case(n=3, xbuf_shape=(21, 21, 64), xview=(1, 2, 19, 18, 0, 64), ybuf_shape=(19, 18, 64), yview=(0, 0, 19, 18, 0, 64),
                      wt=_w(64, 64, 3), pad=(1, 1), bn=True, relu=1, x3=terms)
    _check(got, want, TOL[terms])


@pytest.mark.parametrize("terms", [9, 6])
def test_both_column_tiles_give_the_same_bits(terms):
    kw = dict(n=3, xbuf_shape=(21, 21, 256), xview=(0, 0, 21, 21, 0, 256), ybuf_shape=(21, 21, 256), yview=(0, 0, 21, 21, 0, 256),
              wt=_w(256, 256, 1), bn=True, relu=1, res=True, seed=5, x3=terms)
    g128, want = _case(force_tile=128, **kw)
    g64, _ = _case(force_tile=64, **kw)
    assert torch.equal(g128, g64)
    _check(g128, want, TOL[terms])


@pytest.mark.parametrize("terms", [9, 6])
@pytest.mark.parametrize("cin2,stride2,cout", [(64, 1, 256), (256, 2, 512)])
def test_fused_shortcut(cin2, stride2, cout, terms):
    import plan_interp
    from gpu_util import MiniPlan, rand_conv_weight
    from hover_net_amd import plan as PL
    from hover_net_amd.engine import Engine

    rng = np.random.default_rng(5)
    n, so = 2, 13
    si = (so - 1) * stride2 + 1 + (1 if stride2 == 2 else 0)
    P = MiniPlan()
    x = PL.View(P.buf("t2", so, so, 64))
    x2 = PL.View(P.buf("xin", si, si, cin2))
    y = PL.View(P.buf("y", so, so, cout))
    op = P.conv("fused", x, y, rand_conv_weight(rng, cout, 64, 1), x2=x2, wt2=rand_conv_weight(rng, cout, cin2, 1), stride2=stride2,
                post=(rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.3, cout)))
    op.extra["x3"] = terms
    P.pack()
    eng = Engine(P, max_batch=n, n_split=1)
    eng.arena.copy_(torch.randn(eng.arena.shape, generator=torch.Generator().manual_seed(1)))
    A = plan_interp.Arena(P, n)
    A.flat.copy_(eng.arena.cpu())
    eng.run_raw(n)
    torch.cuda.synchronize()
    want = plan_interp.conv_ref(op, A.view(op.x).clone(), None, A.view(x2).clone())
    _check(eng.buffer(op.y, n).cpu(), want, TOL[terms])


@pytest.mark.parametrize("terms", [9, 6])
@pytest.mark.parametrize("m,k,cin,cout,s,pad", [(4, 5, 1024, 256, 10, (0, 0)), (6, 3, 512, 512, 33, (1, 1)), (4, 5, 256, 64, 12, (2, 2))])
def test_winograd_domain_product(m, k, cin, cout, s, pad, terms):
    """WINO_IN -> the batched transform-domain product on the bf16x3 kernel -> WINO_OUT against a direct fp32 convolution (the
    tolerance of tests/test_gpu_conv.py::test_winograd_5x5_matches_direct for the fp32-MFMA product)."""
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
    P.conv_winograd("w", x, y, wt, pad=pad, m=m)
    gemm = [o for o in P.ops if o.kind == PL.OP_CONV][0]
    gemm.extra["x3"] = terms
    ybuf.first = 0
    P.pack()
    eng = Engine(P, max_batch=n, n_split=1)
    eng.arena.copy_(torch.randn(eng.arena.shape, generator=torch.Generator().manual_seed(3)))
    xin = eng.buffer(x, n).cpu().clone()
    eng.run_raw(n)
    torch.cuda.synchronize()
    want = F.conv2d(F.pad(xin.permute(0, 3, 1, 2), (pad[0], pad[1], pad[0], pad[1])), torch.from_numpy(wt).float()).permute(0, 2, 3, 1)
    _check(eng.buffer(PL.View(ybuf), n).cpu()[..., 32:], want, 5e-4)


# ---- csrc/hvn_conv_x3g.hip: the same convolution with both operands staged by LDS-DMA and the split at the fragment read; tile_n codes
#      896 (256 pixels x 128 channels, 8 waves) and 640 (128 x 128, 4 waves).  Same packing, same k-slot assignment, same MFMA order per
#      accumulator: BIT-IDENTICAL to hvn_conv_x3.hip's 128 x 128 form, for every launch class the plan puts on it. ---------------------------
X3G_FORMS = [896, 640]
X3G_CASES = {
    # name: run_conv_case keywords (M = 338 .. 1323 pixels: partial last tiles of 128 and of 256 rows, tiles straddling samples)
    "1x1_64_256_bn": dict(n=2, xbuf_shape=(13, 13, 64), xview=(0, 0, 13, 13, 0, 64), ybuf_shape=(13, 13, 256), yview=(0, 0, 13, 13, 0, 256), w=(256, 64, 1), bn=True, relu=1),
    "1x1_2048_1024": dict(n=2, xbuf_shape=(13, 13, 2048), xview=(0, 0, 13, 13, 0, 2048), ybuf_shape=(13, 13, 1024), yview=(0, 0, 13, 13, 0, 1024), w=(1024, 2048, 1), bn=True, relu=1),
    "1x1_288_128_window": dict(n=3, xbuf_shape=(21, 21, 320), xview=(1, 2, 19, 18, 0, 288), ybuf_shape=(19, 18, 160), yview=(0, 0, 19, 18, 32, 128), w=(128, 288, 1), bn=True, relu=1),
    "res_post": dict(n=2, xbuf_shape=(17, 17, 64), xview=(0, 0, 17, 17, 0, 64), ybuf_shape=(17, 17, 256), yview=(0, 0, 17, 17, 0, 256), w=(256, 64, 1), res=True, post=True),
    "res_inplace": dict(n=1, xbuf_shape=(20, 20, 64), xview=(0, 0, 20, 20, 0, 64), ybuf_shape=(20, 20, 128), yview=(0, 0, 20, 20, 0, 128), w=(128, 64, 1), res=True, inplace_res=True),
    "prologue_1024_256": dict(n=2, xbuf_shape=(17, 17, 1024), xview=(0, 0, 17, 17, 0, 1024), ybuf_shape=(17, 17, 256), yview=(0, 0, 17, 17, 0, 256), w=(256, 1024, 1), pre=True, bn=True, relu=1),
    "prologue_2048_512": dict(n=1, xbuf_shape=(12, 12, 2048), xview=(0, 0, 12, 12, 0, 2048), ybuf_shape=(12, 12, 512), yview=(0, 0, 12, 12, 0, 512), w=(512, 2048, 1), pre=True, bn=True, relu=1),
    "prologue_3x3_valid": dict(n=2, xbuf_shape=(12, 12, 256), xview=(0, 0, 12, 12, 0, 256), ybuf_shape=(10, 10, 128), yview=(0, 0, 10, 10, 0, 128), w=(128, 256, 3), pre=True, bn=True, relu=1),
    "prologue_res_96_128": dict(n=3, xbuf_shape=(11, 11, 96), xview=(0, 0, 11, 11, 0, 96), ybuf_shape=(11, 11, 128), yview=(0, 0, 11, 11, 0, 128), w=(128, 96, 1), pre=True, res=True),
    "1x1_stride2": dict(n=2, xbuf_shape=(24, 24, 256), xview=(0, 0, 24, 24, 0, 256), ybuf_shape=(12, 12, 512), yview=(0, 0, 12, 12, 0, 512), w=(512, 256, 1), stride=2),
    "3x3_same": dict(n=2, xbuf_shape=(18, 18, 128), xview=(0, 0, 18, 18, 0, 128), ybuf_shape=(18, 18, 128), yview=(0, 0, 18, 18, 0, 128), w=(128, 128, 3), pad=(1, 1), bn=True, relu=1),
    "3x3_same_stride2": dict(n=2, xbuf_shape=(18, 18, 128), xview=(0, 0, 18, 18, 0, 128), ybuf_shape=(9, 9, 128), yview=(0, 0, 9, 9, 0, 128), w=(128, 128, 3), stride=2, pad=(0, 1), bn=True, relu=1),
    "5x5_valid_1024_256": dict(n=1, xbuf_shape=(12, 12, 1024), xview=(0, 0, 12, 12, 0, 1024), ybuf_shape=(8, 8, 512), yview=(0, 0, 8, 8, 0, 256), w=(256, 1024, 5)),
}


def test_device_split_of_the_weight_planes_equals_the_host_split():
    """Engine._upload_params forms the bf16 planes of the fp32 weight packings on the GPU (round 6): bit-equal to engine.pack_conv_x3 (numpy),
    whose arithmetic tests/test_x3_arithmetic.py pins on the CPU -- incl. values across the whole exponent range and exact ties."""
    from hover_net_amd.engine import pack_conv_x3, pack_conv_x3_device
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((3, 64, 5, 9, 32)) * np.exp(rng.uniform(-60, 60, (3, 64, 5, 9, 32)))).astype(np.float32)
    w[0, 0, 0, 0, :8] = np.float32([1.00390625, 1.01171875, -1.00390625, 0.0, -0.0, 2.0 ** -120, 3.0e38, -3.0e38])   # ties to even, zeros, the edges
    got = pack_conv_x3_device(w, "cuda").cpu().numpy().view(np.uint16)
    assert np.array_equal(got, pack_conv_x3(w).ravel())


@pytest.mark.parametrize("form", X3G_FORMS)
@pytest.mark.parametrize("case", sorted(X3G_CASES))
def test_lds_dma_form_gives_the_bits_of_the_staged_form(case, form):
    kw = dict(X3G_CASES[case])
    cout, cin, k = kw.pop("w")
    kw.update(wt=_w(cout, cin, k, seed=3), seed=9, x3=6)
    ref, want = _case(force_tile=128, **kw)
    got, _ = _case(force_tile=form, **kw)
    _check(ref, want, TOL[6])
    assert torch.equal(got, ref), "max abs difference %g" % (got - ref).abs().max().item()


@pytest.mark.parametrize("form", X3G_FORMS)
def test_lds_dma_form_nine_terms_fused_shortcut_and_winograd_product(form):
    import plan_interp
    from gpu_util import MiniPlan, rand_conv_weight
    from hover_net_amd import plan as PL
    from hover_net_amd.engine import Engine

    # nine partial products (the other instantiation family)
    kw = dict(n=2, xbuf_shape=(13, 13, 128), xview=(0, 0, 13, 13, 0, 128), ybuf_shape=(13, 13, 512), yview=(0, 0, 13, 13, 0, 512),
              wt=_w(512, 128, 1), bn=True, relu=1, res=True, seed=4, x3=9)
    ref, want = _case(force_tile=128, **kw)
    got, _ = _case(force_tile=form, **kw)
    _check(ref, want, TOL[9])
    assert torch.equal(got, ref)
    # the strided 1x1 shortcut as a second K source (d1 .. d3 unit 0)
    rng = np.random.default_rng(5)
    outs = []
    for tile in (128, form):
        P = MiniPlan()
        x = PL.View(P.buf("t2", 13, 13, 64))
        x2 = PL.View(P.buf("xin", 26, 26, 256))
        y = PL.View(P.buf("y", 13, 13, 512))
        rng = np.random.default_rng(5)
        op = P.conv("fused", x, y, rand_conv_weight(rng, 512, 64, 1), x2=x2, wt2=rand_conv_weight(rng, 512, 256, 1), stride2=2,
                    post=(rng.uniform(0.5, 1.5, 512), rng.normal(0, 0.3, 512)))
        op.extra["x3"] = 6
        P.pack()
        eng = Engine(P, max_batch=2, n_split=1)
        eng.arena.copy_(torch.randn(eng.arena.shape, generator=torch.Generator().manual_seed(1)))
        eng.ops[0].tile_n = tile
        if tile == 128:
            A = plan_interp.Arena(P, 2)
            A.flat.copy_(eng.arena.cpu())
            want = plan_interp.conv_ref(op, A.view(op.x).clone(), None, A.view(x2).clone())
        eng.run_raw(2)
        torch.cuda.synchronize()
        outs.append(eng.buffer(op.y, 2).cpu().clone())
    _check(outs[0], want, TOL[6])
    assert torch.equal(outs[0], outs[1])
    # the batched transform-domain product of a Winograd convolution (blockIdx.y = position)
    outs = []
    for tile in (128, form):
        P = MiniPlan()
        x = PL.View(P.buf("x", 33, 33, 512))
        ybuf = P.buf("y", 33, 33, 512)
        P.conv_winograd("w", x, PL.View(ybuf), rand_conv_weight(np.random.default_rng(7), 512, 512, 3), pad=(1, 1), m=6)
        gi = [i for i, o in enumerate(P.ops) if o.kind == PL.OP_CONV][0]
        P.ops[gi].extra["x3"] = 6
        ybuf.first = 0
        P.pack()
        eng = Engine(P, max_batch=2, n_split=1)
        eng.arena.copy_(torch.randn(eng.arena.shape, generator=torch.Generator().manual_seed(3)))
        eng.ops[gi].tile_n = tile
        eng.run_raw(2)
        torch.cuda.synchronize()
        outs.append(eng.buffer(PL.View(ybuf), 2).cpu().clone())
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("name", ["orig5", "fast6"])
def test_network_with_the_lds_dma_forms_forced_is_bit_equal(name, monkeypatch):
    """Every bf16x3 launch that has an LDS-DMA form on that form (HVN_X3G_FORCE = 896 | 640) against hvn_conv_x3.hip everywhere (HVN_X3G=0):
    the logits carry the same bits -- which form a launch runs on is a timing decision of the engine, invisible in the results."""
    from test_oracle_net import load_case
    from hover_net_amd import net_desc, plan as PL, run_desc

    mode, nt, sd, tiles, crop, logits, pmap = load_case(name)
    x = torch.from_numpy(tiles)
    outs, counts = [], []
    for env in ({"HVN_X3G": "0"}, {"HVN_X3G_FORCE": "896"}, {"HVN_X3G_FORCE": "640"}):
        for k in ("HVN_X3G", "HVN_X3G_FORCE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3)
        net.load_state_dict(sd, strict=True)
        net = net.to("cuda").eval()
        run_desc.infer_step_device(x, net)
        eng = net.engine(x.shape[0])
        counts.append(sum(1 for o in eng.ops if o.kind == PL.OP_CONV and o.tile_n in (896, 640)))
        outs.append({k: eng.logits[k][:x.shape[0]].cpu().clone() for k in eng.logits})
    assert counts[0] == 0 and counts[1] > 40 and counts[2] > 40, counts
    for o in outs[1:]:
        for k in outs[0]:
            assert torch.equal(outs[0][k], o[k]), k


@pytest.mark.parametrize("terms", ["9", "6"])
@pytest.mark.parametrize("name", ["orig5", "fast6"])
def test_network_on_the_bf16x3_kernels_matches_reference_golden(name, terms, monkeypatch):
    """The whole network with every MFMA-bound launch on csrc/hvn_conv_x3.hip (HVN_X3 = 9 | 6) against the goldens made by the reference's
    own code: 1e-3 on the logits; and a tile alone gives the bits it gives inside a batch (static kernel choice, not a timing)."""
    from test_oracle_net import crop_to, load_case
    from hover_net_amd import net_desc, plan as PL, run_desc

    monkeypatch.setenv("HVN_X3", terms)
    mode, nt, sd, tiles, crop, logits, pmap = load_case(name)
    net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3)
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda").eval()
    x = torch.from_numpy(tiles)
    got = run_desc.infer_step_device(x, net).cpu().numpy().copy()
    eng = net.engine(x.shape[0])
    n_x3 = sum(1 for o in eng.plan.ops if o.kind in (PL.OP_CONV, PL.OP_CHAIN) and o.extra.get("x3"))      # + d0's chained seams (round 5)
    assert n_x3 > 50 and sum(1 for o in eng.ops if o.act_dtype in (2, 3)) == n_x3
    worst = 0.0
    for k, v in logits.items():
        g = crop_to(eng.logits[k][:x.shape[0]].cpu().numpy(), crop, (2, 3))
        worst = max(worst, float(np.abs(g - v).max()))
    print("%s HVN_X3=%s: %d launches on the bf16x3 kernel, max |logit - golden| %.2e" % (name, terms, n_x3, worst))
    assert worst <= 1e-3
    from hover_net_amd.synth import synth_tiles
    more = torch.from_numpy(synth_tiles(5, x.shape[1], seed=77))
    batch = run_desc.infer_step_device(more, net).cpu().numpy().copy()
    alone = run_desc.infer_step_device(more[3:4], net).cpu().numpy()
    assert np.array_equal(alone[0], batch[3])

"""fp32 parity at a REALISTIC activation scale (round-3 verdict, missing #3 / next #3).

Every other fp32 logit test runs `synth_state_dict`, whose residual convs are damped so that activations stay O(1..10).  A real
pre-activation ResNet-50 checkpoint is not like that: a block's output is an UN-normalised running sum over its units
(/root/reference/models/hovernet/net_utils.py:250-266) and Winograd's fp32 error scales with the activation range.  No checkpoint
can be downloaded, so one is FITTED here with the repository's own trainer (hover_net_amd/synth_fit.py) from the reference's own
initialisation (`Net.weights_init`: un-damped Kaiming convs, BatchNorm 1 / 0) -- trained BatchNorm statistics, a network that
segments -- and then, on held-out painted tiles:

  * HIP fp32 logits vs the torch-CPU fp32 oracle (oracle/net_torch.py) within 1e-3 (BASELINE north_star), for the default lowering
    (products on the bf16 pipe from bf16x3 splits, 6 terms; F(6x6,3x3) encoder; F(6x6,5x5) u3, F(4x4,5x5) u2 / u1) and for each
    alternative: every conv on the fp32 matrix pipe, direct convolutions on either pipe, 9 terms, F(4x4) tiles, chained d1 -- and,
    reported but not asserted, the F(6x6,5x5) options that are not shipped because they spend the margin;
  * the on-GPU instance maps of those NETWORK outputs vs the C oracle (oracle/hvn_oracle.c) bit for bit
    (/root/reference/models/hovernet/run_desc.py:171-197 -> post_proc.py:27-90).

max |activation| per stage and the measured error / margin are printed (`-s`), and quoted in DESIGN.md."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.fitted]

TOL = 1e-3
# (HVN_WINOGRAD, HVN_WINOGRAD3_M, HVN_WINOGRAD3): decoder F(m,5) tile | encoder F(m,3) tile | 0 = no Winograd at all
LOWERINGS = [("default", {}),                                                     # bf16x3 6-term products, F(6,3) d1 / d2, F(6,5) u3, F(4,5) u2 / u1
             ("fp32 matrix pipe", {"HVN_X3": "0"}),                               # round 3's arithmetic on this round's Winograd tiles
             ("direct convolutions, fp32 pipe", {"HVN_X3": "0", "HVN_WINOGRAD": "0"}),
             ("direct convolutions, bf16x3", {"HVN_WINOGRAD": "0"}),
             ("bf16x3, 9 terms", {"HVN_X3": "9"}),
             ("conservative (HoVerNet.lowering)", {"lowering": "conservative"}),   # the env-free switch: 9 terms + F(4x4, .) tiles everywhere
             ("d0's seams chained on the fp32 pipe", {"HVN_X3_CHAIN": ""}),        # rounds 3-4's d0
             ("F(4,5) for u3 as well", {"HVN_WINOGRAD_STAGES": "none:0"}),
             ("F(4,3) encoder", {"HVN_WINOGRAD3_M": "4"}),
             ("d1's 1x1 convs chained on the fp32 pipe", {"HVN_X3_D1": "0"}),
             ("F(6,5) u3 + u2", {"HVN_WINOGRAD_STAGES": "u3:6,u2:6"}),                # measured options that are NOT shipped: reported, not asserted
             ("F(6,5) u3 + u1", {"HVN_WINOGRAD_STAGES": "u3:6,u1:6"}),
             ("F(6,5) everywhere", {"HVN_WINOGRAD": "6"})]


# the default run checks the shipped lowering and the four it is judged against; HVN_TRAINED_LIKE_FULL=1 adds every alternative (that is
# how profiles/r0*_trained_like_margins.txt were made)
CORE = ("default", "fp32 matrix pipe", "direct convolutions, fp32 pipe", "conservative (HoVerNet.lowering)", "F(6,5) everywhere")
ENV_KEYS = ("HVN_WINOGRAD", "HVN_WINOGRAD3_M", "HVN_WINOGRAD3", "HVN_WINOGRAD_STAGES", "HVN_X3", "HVN_X3_D1", "HVN_X3_CHAIN")

_FITS = {}


def _fitted(mode, nr_types):
    """The fitted checkpoint of (mode, nr_types), made once per test process: (state_dict on the host, loss curve, init name)."""
    import fit_util

    key = (mode, nr_types)
    if key not in _FITS:
        size = 270 if mode == "original" else 256
        init = os.environ.get("HVN_FIT_INIT", "kaiming")
        tnet, curve = fit_util.fit(mode, nr_types, steps=int(os.environ.get("HVN_FIT_STEPS", "240")), lr=1e-3, seed=0, init=init,
                                   density=fit_util.consep_density(size))
        assert np.mean(curve[-30:]) < 0.7 * np.mean(curve[10:40]), "the fit did not converge: %s" % curve[::40]
        sd = {k: v.detach().cpu().clone() for k, v in tnet.state_dict().items()}
        tnet._train_engine = None
        del tnet
        torch.cuda.empty_cache()
        _FITS[key] = (sd, curve, init)
    return _FITS[key]


def _hip_logits(sd, mode, nr_types, tiles, env, monkeypatch, lowering="default"):
    from hover_net_amd import net_desc, run_desc

    for k in ENV_KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    net = net_desc.create_model(mode=mode, nr_types=nr_types, input_ch=3)
    net.load_state_dict(sd, strict=True)
    net.max_batch = tiles.shape[0]
    net.lowering = lowering
    net = net.to("cuda").eval()
    pred = run_desc.infer_step_device(tiles, net).clone()
    eng = net.engine(tiles.shape[0])
    logits = {k: eng.logits[k][:tiles.shape[0]].cpu().clone() for k in eng.logits}
    del net, eng
    torch.cuda.empty_cache()
    return logits, pred


@pytest.mark.parametrize("mode,nr_types", [("original", 5), ("fast", 6)])
def test_fp32_parity_on_a_trained_like_checkpoint(mode, nr_types, monkeypatch):
    import fit_util
    from hover_net_amd import net_desc, post_proc, run_desc
    from oracle import net_torch
    from oracle import postproc as O
    from pq_util import pq

    size, out = (270, 80) if mode == "original" else (256, 164)
    dens = fit_util.consep_density(size)
    sd, curve, init = _fitted(mode, nr_types)

    n = 4
    imgs, anns, _typs = fit_util.painted_tiles(n, size, seed=4242, k_lo=dens[0], k_hi=dens[1], nr_types=nr_types)
    tiles = torch.from_numpy(imgs)
    taps = {}
    want = net_torch.forward(sd, tiles.permute(0, 3, 1, 2).float(), mode, taps=taps)
    want_pm = net_torch.infer_epilogue(want)
    scale = {k: float(v.abs().max()) for k, v in taps.items()}
    print("\n%s/%d trained-like (init %s): loss %.3f -> %.3f; max |activation| per stage: %s; max |logit| %s"
          % (mode, nr_types, init, np.mean(curve[:10]), np.mean(curve[-10:]), " ".join("%s %.1f" % kv for kv in sorted(scale.items())),
             " ".join("%s %.1f" % (k, float(v.abs().max())) for k, v in want.items())))
    o = (size - out) // 2
    truth = anns[:, o:o + out, o:o + out]
    errs = {}
    full = os.environ.get("HVN_TRAINED_LIKE_FULL", "0") != "0"
    for name, env in LOWERINGS:
        if not full and name not in CORE:
            continue
        if mode == "fast" and "HVN_WINOGRAD_STAGES" in env:
            continue                       # 'fast' mode has no 5x5 convs
        env = dict(env)
        got, pred = _hip_logits(sd, mode, nr_types, tiles, env, monkeypatch, lowering=env.pop("lowering", "default"))
        err = max(float((got[k] - want[k]).abs().max()) for k in want)
        errs[name] = err
        # the instance separation of the NETWORK's output: GPU vs the C oracle on the same map, bit for bit
        inst, _, counts = post_proc.process_batch_device(pred, nr_types=nr_types)
        pm = pred.cpu().numpy()
        np.testing.assert_array_equal(inst.cpu().numpy(), O.proc_batch(pm))
        if name == "default":
            assert int(counts.sum().item()) > 0, "the fitted network emits no nuclei"
            q = [pq(truth[i], inst[i].cpu().numpy()) for i in range(n)]
            c0 = 1
            dp = float((pred.cpu()[..., c0] - want_pm[..., c0]).abs().max())
            print("  instances %d on %d tiles, PQ vs the painted truth %.3f, max |p_nuc - oracle| %.2e" % (int(counts.sum().item()), n, float(np.mean(q)), dp))
            assert np.mean(q) > 0.5, "the fitted network does not segment: PQ vs truth %.3f" % np.mean(q)
    print("  max |logit - oracle|: " + "; ".join("%s %.2e (margin %.0fx)" % (k, v, TOL / max(v, 1e-12)) for k, v in errs.items()))
    for name, err in errs.items():
        if name.startswith("F(6,5)"):
            continue                       # measured options that are NOT shipped (reported above: they spend the margin)
        assert err <= TOL, (name, err)


def test_parity_margin_against_activation_scale(monkeypatch):
    """Where does the margin end?  The fitted 'original' checkpoint made HOTTER: gamma and beta of the block-closing BatchNorms of d1, d2, d3
    (net_utils.py:262-266 -- what every later stage and every skip connection is fed with) scaled by s = 2, 4.  For each: max
    |activation| and max |logit| of the fp32 oracle, and max |logit - oracle| of the shipped lowering, of the fp32 matrix pipe and of
    `HoVerNet.lowering = "conservative"` (direct convolutions on the fp32 pipe, measured once: 3.1e-3 / 3.4e-2, profiles/r05_trained_like_margins.txt).  Two fp32 evaluations of a hotter network differ by more
    (the error is relative to the activations, the 1e-3 of BASELINE north_star is absolute), so what is asserted is that the shipped and
    the conservative lowering stay within max(1e-3, 2 x the fp32 pipe's own distance from the oracle); the table (`-s`) is what
    DESIGN.md section 2 quotes for the logit magnitude at which the default leaves 1e-3."""
    import fit_util
    from oracle import net_torch

    mode, nr_types, size = "original", 5, 270
    sd0, _curve, _init = _fitted(mode, nr_types)
    dens = fit_util.consep_density(size)
    imgs = fit_util.painted_tiles(2, size, seed=777, k_lo=dens[0], k_hi=dens[1], nr_types=nr_types)[0]
    tiles = torch.from_numpy(imgs)
    rows = []
    for s in (2.0, 4.0):                 # (x1 is the test above: default 2.1e-4 .. 2.9e-4)
        sd = {k: v.clone() for k, v in sd0.items()}
        for blk in ("d1", "d2", "d3"):
            for leaf in ("weight", "bias"):
                sd["%s.blk_bna.bn.%s" % (blk, leaf)] = sd["%s.blk_bna.bn.%s" % (blk, leaf)] * s
        taps = {}
        want = net_torch.forward(sd, tiles.permute(0, 3, 1, 2).float(), mode, taps=taps)
        act = max(float(v.abs().max()) for v in taps.values())
        logit = max(float(v.abs().max()) for v in want.values())
        errs = {}
        for name, env, low in (("default", {}, "default"), ("fp32 pipe", {"HVN_X3": "0"}, "default"), ("conservative", {}, "conservative")):
            got, _ = _hip_logits(sd, mode, nr_types, tiles, env, monkeypatch, lowering=low)
            errs[name] = max(float((got[k] - want[k]).abs().max()) for k in want)
        rows.append((s, act, logit, errs))
        print("\n  scale x%g: max |activation| %.0f, max |logit| %.0f; max |logit - oracle|: %s"
              % (s, act, logit, "; ".join("%s %.2e" % kv for kv in errs.items())))
    for s, act, logit, errs in rows:
        bound = max(TOL, 2.0 * errs["fp32 pipe"])
        assert errs["default"] <= bound and errs["conservative"] <= bound, (s, errs)

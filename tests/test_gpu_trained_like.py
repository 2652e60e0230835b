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

pytestmark = pytest.mark.gpu

TOL = 1e-3
# (HVN_WINOGRAD, HVN_WINOGRAD3_M, HVN_WINOGRAD3): decoder F(m,5) tile | encoder F(m,3) tile | 0 = no Winograd at all
LOWERINGS = [("default", {}),                                                     # bf16x3 6-term products, F(6,3) d1 / d2, F(6,5) u3, F(4,5) u2 / u1
             ("fp32 matrix pipe", {"HVN_X3": "0"}),                               # round 3's arithmetic on this round's Winograd tiles
             ("direct convolutions, fp32 pipe", {"HVN_X3": "0", "HVN_WINOGRAD": "0"}),
             ("direct convolutions, bf16x3", {"HVN_WINOGRAD": "0"}),
             ("bf16x3, 9 terms", {"HVN_X3": "9"}),
             ("F(4,5) for u3 as well", {"HVN_WINOGRAD_STAGES": "none:0"}),
             ("F(4,3) encoder", {"HVN_WINOGRAD3_M": "4"}),
             ("d1's 1x1 convs chained on the fp32 pipe", {"HVN_X3_D1": "0"}),
             ("F(6,5) u3 + u2", {"HVN_WINOGRAD_STAGES": "u3:6,u2:6"}),                # measured options that are NOT shipped: reported, not asserted
             ("F(6,5) u3 + u1", {"HVN_WINOGRAD_STAGES": "u3:6,u1:6"}),
             ("F(6,5) everywhere", {"HVN_WINOGRAD": "6"})]


# the default run checks the shipped lowering and the four it is judged against; HVN_TRAINED_LIKE_FULL=1 adds every alternative (that is
# how profiles/r04_trained_like_margins.txt was made)
CORE = ("default", "fp32 matrix pipe", "direct convolutions, fp32 pipe", "bf16x3, 9 terms", "F(6,5) everywhere")


@pytest.mark.parametrize("mode,nr_types", [("original", 5), ("fast", 6)])
def test_fp32_parity_on_a_trained_like_checkpoint(mode, nr_types, monkeypatch):
    import fit_util
    from hover_net_amd import net_desc, post_proc, run_desc
    from oracle import net_torch
    from oracle import postproc as O
    from pq_util import pq

    size, out = (270, 80) if mode == "original" else (256, 164)
    dens = fit_util.consep_density(size)
    init = os.environ.get("HVN_FIT_INIT", "kaiming")
    tnet, curve = fit_util.fit(mode, nr_types, steps=int(os.environ.get("HVN_FIT_STEPS", "240")), lr=1e-3, seed=0, init=init, density=dens)
    assert np.mean(curve[-30:]) < 0.7 * np.mean(curve[10:40]), "the fit did not converge: %s" % curve[::40]
    sd = {k: v.detach().cpu().clone() for k, v in tnet.state_dict().items()}
    tnet._train_engine = None
    del tnet
    torch.cuda.empty_cache()

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
        for k in ("HVN_WINOGRAD", "HVN_WINOGRAD3_M", "HVN_WINOGRAD3", "HVN_WINOGRAD_STAGES", "HVN_X3", "HVN_X3_D1"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = net_desc.create_model(mode=mode, nr_types=nr_types, input_ch=3)
        net.load_state_dict(sd, strict=True)
        net.max_batch = n
        net = net.to("cuda").eval()
        pred = run_desc.infer_step_device(tiles, net).clone()
        eng = net.engine(n)
        err = max(float((eng.logits[k][:n].cpu() - want[k]).abs().max()) for k in want)
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
        del net, eng
        torch.cuda.empty_cache()
    print("  max |logit - oracle|: " + "; ".join("%s %.2e (margin %.0fx)" % (k, v, TOL / max(v, 1e-12)) for k, v in errs.items()))
    for name, err in errs.items():
        if name.startswith("F(6,5)"):
            continue                       # measured options that are NOT shipped (reported above: they spend the margin)
        assert err <= TOL, (name, err)

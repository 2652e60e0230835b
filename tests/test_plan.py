"""CPU: the lowering (hover_net_amd/plan.py) interpreted with torch ops equals the oracle --
BN folding, concat-by-offset, crops, block-diagonal grouped convs, arena packing -- and the
host-side mirror of the reference interface keeps its contract."""
import numpy as np
import pytest
import torch

import plan_interp
from hover_net_amd import arch
from hover_net_amd import plan as PL
from hover_net_amd.synth import synth_state_dict, synth_tiles
from oracle import net_torch


@pytest.mark.parametrize("mode,nt,gflop", [("original", 5, 392.17), ("fast", 6, 297.87), ("original", None, 322.22)])
def test_plan_matches_oracle(mode, nt, gflop):
    sd = synth_state_dict(mode, nt, seed=3)
    P = PL.build_plan(sd, mode, nt)
    assert abs(P.total_flops() / 1e9 - gflop) < 0.01  # SURVEY.md 2.2(i) totals
    imgs = torch.from_numpy(synth_tiles(1, P.geo["inp"], seed=5))
    ref = net_torch.forward(sd, imgs.permute(0, 3, 1, 2).float(), mode)
    got, pred = plan_interp.run(P, imgs)
    for k in ref:
        assert float((ref[k] - got[k]).abs().max()) < 1e-4
    pr = net_torch.infer_epilogue(ref)
    assert float((pr[..., -3:] - pred[..., -3:]).abs().max()) < 1e-4


def test_arena_packing_has_no_live_overlap():
    P = PL.build_plan(synth_state_dict("original", 5, seed=1), "original", 5)
    live = [b for b in P.bufs if b.last >= 0]
    for i, a in enumerate(live):
        assert a.offset % 64 == 0 and a.offset + a.size <= P.arena_per_sample
        for b in live[i + 1:]:
            if a.last < b.first or b.last < a.first:
                continue
            assert a.offset + a.size <= b.offset or b.offset + b.size <= a.offset, (a.name, b.name)
    assert P.arena_per_sample * 4 < 600e6  # ~555 MB / tile: the Winograd F(4x4,5x5) transform-domain tensors (64/16 x the 5x5 convs' inputs and outputs) of three concurrent branches


def test_every_conv_is_kernel_legal():
    for mode, nt in (("original", 5), ("fast", 6)):
        P = PL.build_plan(synth_state_dict(mode, nt, seed=1), mode, nt)
        for op in P.ops:
            if op.kind != PL.OP_CONV:
                continue
            assert op.x.c % 32 == 0 and op.x.c0 % 4 == 0 and op.y.c0 % 4 == 0, op.name
            x2 = op.extra.get('x2')
            w = op.w[0] if op.extra.get("nbatch", 1) > 1 else op.w      # batched Winograd GEMM: [n*n] stacked weight sets
            assert w.shape[0] % op.tile_n == 0 and w.shape[1] * 32 == op.x.c + (x2.c if x2 is not None else 0) and w.shape[3] == 32
            assert op.tile_n in (32, 64, 128)


def test_param_table_counts():
    # 632 keys seg-only / 798 with the tp branch; 45.03 M / 54.74 M / 37.64 M params
    def count(t):
        return sum(int(np.prod(s)) for k, (kind, s) in t.items() if kind in ("conv", "bias", "bn_w", "bn_b"))

    assert len(arch.param_table("original", None)) == 632
    assert len(arch.param_table("original", 5)) == 798
    assert abs(count(arch.param_table("original", None)) / 1e6 - 45.03) < 0.01
    assert abs(count(arch.param_table("original", 5)) / 1e6 - 54.74) < 0.01
    assert abs(count(arch.param_table("fast", 6)) / 1e6 - 37.64) < 0.01


def test_module_contract_cpu():
    from hover_net_amd import net_desc

    net = net_desc.create_model(mode="original", nr_types=5, input_ch=3)
    assert (net.mode, net.freeze, net.nr_types, net.output_ch) == ("original", False, 5, 4)
    sd = net.state_dict()
    assert list(sd.keys()) == list(arch.param_table("original", 5).keys())
    assert sd["conv0./.weight"].shape == (64, 3, 7, 7) and sd["upsample2x.unpool_mat"].shape == (2, 2)
    assert sd["d0.units.1.preact/bn.num_batches_tracked"].dtype == torch.long
    net.load_state_dict(synth_state_dict("original", 5, seed=2), strict=True)
    with pytest.raises(RuntimeError):  # missing key must fail like the reference's strict load
        bad = dict(synth_state_dict("original", 5, seed=2))
        bad.pop("conv_bot.weight")
        net.load_state_dict(bad, strict=True)
    net.eval()
    with pytest.raises(RuntimeError):  # no CPU fallback
        net(torch.zeros(1, 3, 270, 270))
    with pytest.raises(AssertionError):
        net_desc.HoVerNet(mode="bogus")
    seg = net_desc.create_model(mode="fast", nr_types=None)
    assert seg.output_ch == 3 and len(seg.state_dict()) == 632


def test_abi_exports_every_declared_symbol():
    import re
    import os
    from hover_net_amd import lib as L

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "hvn.h")).read()
    declared = set(re.findall(r"HVN_API\s+[\w\s\*]+?\b(hvn_\w+)\s*\(", hdr))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hvn_version() == 104


@pytest.mark.parametrize("env,tol", [({"HVN_CHAIN": "0"}, 1e-4), ({"HVN_FUSE_UPADD": "1"}, 1e-4), ({"HVN_WINOGRAD3_M": "4"}, 1e-4),
                                     ({"HVN_WINOGRAD": "6", "HVN_WINOGRAD3_M": "6"}, 5e-4), ({"HVN_WINOGRAD": "0"}, 1e-4)])
def test_plan_options_match_oracle(env, tol, monkeypatch):
    """The lowering options next to the default -- chains off, UPADD fused into the Winograd input transform, F(4x4,3x3) (the default is F(6x6,3x3) since round 4) /
    F(6x6,5x5) Winograd tiles, no Winograd at all -- interpreted with torch ops equal the oracle (F(6x6,5x5): ten interpolation
    points, 2e-4 on the logits, which is why it is not the default)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sd = synth_state_dict("original", 5, seed=3)
    P = PL.build_plan(sd, "original", 5)
    assert abs(P.total_flops() / 1e9 - 392.17) < 0.01
    imgs = torch.from_numpy(synth_tiles(1, P.geo["inp"], seed=5))
    ref = net_torch.forward(sd, imgs.permute(0, 3, 1, 2).float(), "original")
    got, _ = plan_interp.run(P, imgs)
    for k in ref:
        assert float((ref[k] - got[k]).abs().max()) < tol, (env, k)


def test_bf16_plan_chains_the_seams_its_kernel_has_and_static_wgrad_rule():
    """Round 6 host logic.  The bf16 plan (`build_plan(chain="bf16:...")`, what `HoVerNet.compute_dtype = "bf16"` builds) chains exactly the conv3 ->
    conv1 seams csrc/hvn_conv_chain_bf16.hip has a form for -- conv3's reduction 64 or 128 channels in whole 64-channel slabs, cout % 256 == 0,
    cout2 <= 128: d0's three and d1's two plain-residual ones ('fast' mode: d1 unit 0's fused shortcut has K = 384, d1 -> d2 has cout2 = 256) -- the
    chained plan is the same function (torch interpreter vs the oracle), and the deterministic weight-gradient split is a function of the
    launch shape only."""
    from hover_net_amd import train_engine as TE
    sd = synth_state_dict("fast", 6, seed=3)
    names = {}
    for chain in ("bf16:d0", "bf16:d0d1"):
        P = PL.build_plan(sd, "fast", 6, winograd=0, chain=chain, x3=0)
        names[chain] = [o.name for o in P.ops if o.kind == PL.OP_CHAIN]
    assert names["bf16:d0"] == ["d0.units.0.conv3+units.1.conv1", "d0.units.1.conv3+units.2.conv1", "d0.units.2.conv3+d1.units.0.conv1"]
    assert names["bf16:d0d1"] == names["bf16:d0"] + ["d1.units.1.conv3+units.2.conv1", "d1.units.2.conv3+units.3.conv1"]
    for o in P.ops:
        if o.kind == PL.OP_CHAIN:
            x2 = o.extra.get("x2")
            assert o.x.c % 64 == 0 and o.x.c + (x2.c if x2 is not None else 0) in (64, 128) and o.cout % 256 == 0 and o.extra["cout2"] in (64, 128)
    imgs = torch.from_numpy(synth_tiles(1, P.geo["inp"], seed=5))
    ref = net_torch.forward(sd, imgs.permute(0, 3, 1, 2).float(), "fast")
    got, _ = plan_interp.run(P, imgs)
    for k in ref:
        assert float((ref[k] - got[k]).abs().max()) < 1e-4, k
    assert TE.static_wgrad_target(1, 1) == 768 and TE.static_wgrad_target(3, 3) == TE.static_wgrad_target(5, 5) == 1024


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """Build hygiene (round-2 verdict, weak #10): the library carries the id of the sources it was compiled from; the binding rebuilds
    by id (not by mtime) and refuses to load a binary that does not match the sources next to it."""
    from hover_net_amd import lib

    assert lib._built_id(lib.LIB_PATH) == lib.source_id()
    assert lib.lib().hvn_build_id().decode() == lib.source_id()
    monkeypatch.setattr(lib, "SOURCES", tuple(lib.SOURCES[:-1]))        # "the sources changed"
    monkeypatch.setattr(lib, "_LIB", None)
    with pytest.raises(lib.HvnError, match="built from other sources"):
        lib.lib()


@pytest.mark.parametrize("mode,nt", [("original", 5), ("fast", 6), ("original", None)])
def test_every_conv_launch_stays_inside_the_kernels_32bit_reach(mode, nt):
    """The conv kernels address a tile's rows with 32-bit byte offsets relative to the sample of its first row (2^31 and beyond is the
    descriptor's "load zeros" range).  A 128-row tile reaches (HoWo + 126) // HoWo samples ahead: one for every spatial conv, but FOUR
    for a Winograd-domain product with 36 tiles per sample -- at the arena's sample stride that left the reach ('fast' mode d3 under
    F(6x6,3x3), round 4: samples 31 and 63 of a batch of 64 read zeros).  `Plan.conv_winograd` keeps F(4x4) for such layers and the
    launchers refuse a violating launch; this pins the plan side for the default lowering of every configuration."""
    P = PL.build_plan(synth_state_dict(mode, nt, seed=0), mode, nt)
    for o in P.ops:
        if o.kind not in (PL.OP_CONV, PL.OP_CHAIN):
            continue
        howo = o.y.h * o.y.w
        ahead = (howo + 126) // howo
        assert ahead * P.arena_per_sample * 4 < 2 ** 31, (o.name, howo, ahead, P.arena_per_sample)
    d3 = [o for o in P.ops if o.kind == PL.OP_WINO_IN and o.name.startswith("d3.")]
    assert d3 and all(o.extra["m"] == 4 for o in d3)
    d2 = [o for o in P.ops if o.kind == PL.OP_WINO_IN and o.name.startswith("d2.")]
    assert d2 and all(o.extra["m"] == 6 for o in d2)


def test_bf16x3_lowering_rule_and_weight_planes(monkeypatch):
    """Which launches form their products on the bf16 pipe is a STATIC rule by layer (`Plan.mark_x3`): every dense conv with a 128- /
    64-wide column tile except d0's very first 1x1; since round 5 d0's seams (conv3 -> next conv1, the last one into d1's first conv1)
    are bf16x3 too and run CHAINED on csrc/hvn_conv_chain_x3.hip (HVN_X3_CHAIN="" keeps rounds 3-4's fp32-pipe chains); d1's other 1x1
    convs run unchained.  The chained and the unchained lowering mark the same layers; HVN_X3=0 marks none.  And the three bf16 planes
    the engine uploads for such a launch sum back to the fp32 weights EXACTLY (h + m + l == w in fp32 arithmetic), in the k-step order
    the kernel walks."""
    import re

    from hover_net_amd.engine import pack_conv_x3, split_bf16x3

    sd = synth_state_dict("original", 5, seed=3)
    P = PL.build_plan(sd, "original", 5)
    x3 = [o for o in P.ops if o.kind == PL.OP_CONV and o.extra.get("x3")]
    assert len(x3) == 93 and all(o.extra["x3"] == 6 for o in x3)
    chains = [o for o in P.ops if o.kind == PL.OP_CHAIN]
    assert [o.name for o in chains] == ["d0.units.0.conv3+units.1.conv1", "d0.units.1.conv3+units.2.conv1", "d0.units.2.conv3+d1.units.0.conv1"]
    assert all(o.extra.get("x3") == 6 for o in chains)                    # both GEMMs of a seam on the bf16 pipe
    assert not any(re.match(r"^d0\.units\.\d+\.conv[13]$", o.name) or o.name == "d1.units.0.conv1" for o in x3)      # (they are inside the chains)
    assert not any(int(o.extra.get("groups", 1)) != 1 for o in x3)
    monkeypatch.setenv("HVN_CHAIN", "0")
    P0 = PL.build_plan(sd, "original", 5)
    parts = sorted(p_.name for o in chains for p_ in o.extra["parts"])
    assert sorted(o.name for o in P0.ops if o.extra.get("x3")) == sorted([o.name for o in x3] + parts)   # the rule does not depend on the chain pass
    monkeypatch.delenv("HVN_CHAIN")
    monkeypatch.setenv("HVN_X3_CHAIN", "")                               # rounds 3-4: d0's seams chained on the fp32 pipe
    Pf = PL.build_plan(sd, "original", 5)
    assert sorted(o.name for o in Pf.ops if o.kind == PL.OP_CONV and o.extra.get("x3")) == sorted(o.name for o in x3)
    assert [bool(o.extra.get("x3")) for o in Pf.ops if o.kind == PL.OP_CHAIN] == [False] * 3
    monkeypatch.delenv("HVN_X3_CHAIN")
    monkeypatch.setenv("HVN_X3", "0")
    assert not any(o.extra.get("x3") for o in PL.build_plan(sd, "original", 5).ops)
    # the planes: exact three-way split, and the packing's layout [cout_pad][k-step][3][32]
    op = next(o for o in x3 if o.name == "d2.units.0.conv2")
    h, m, l = split_bf16x3(op.w)
    f = lambda b: (b.astype(np.uint32) << 16).view(np.float32)     # noqa: E731
    assert np.array_equal((f(h) + f(m)) + f(l), op.w)                                # exact in fp32 arithmetic, not just in float64
    assert np.abs(f(m)).max() <= np.abs(op.w).max() * 2.0 ** -8 and np.abs(f(l)).max() <= np.abs(op.w).max() * 2.0 ** -16
    pk = pack_conv_x3(op.w)
    cout_pad, slabs, taps, _ = op.w.shape
    assert pk.shape == (cout_pad, slabs * taps, 3, 32) and pk.dtype == np.uint16
    assert np.array_equal(pk[:, :, 0, :].reshape(op.w.shape), h) and np.array_equal(pk[:, :, 2, :].reshape(op.w.shape), l)
    # the Winograd-domain product keeps its leading position axis
    g = next(o for o in x3 if o.name.endswith("u3.conva.wino_gemm"))
    assert pack_conv_x3(g.w).shape == g.w.shape[:2] + (g.w.shape[2] * g.w.shape[3], 3, 32)

"""The training lowering (hover_net_amd.train_plan: forward + backward op lists, shared / dilated gradient
buffers, freeze scoping) interpreted on the CPU in float64 must reproduce the training oracle's autograd
gradients to rounding (the oracle itself is pinned to the reference in test_oracle_train.py)."""
import pytest
import torch

import train_interp
from hover_net_amd.synth import synth_state_dict, synth_train_batch
from hover_net_amd.train_plan import TrainPlan
from oracle import train_torch


@pytest.mark.parametrize("mode,nt,freeze", [("original", 5, True), ("fast", None, False)])
def test_lowering_matches_oracle_fp64(mode, nt, freeze):
    torch.set_num_threads(8)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        sd = synth_state_dict(mode, nt, seed=3)
        batch = synth_train_batch(1, mode, nt, seed=11)
        ref = train_torch.train_step(sd, batch, mode, nt, freeze, dtype=torch.float64)
        P = TrainPlan(mode, nt, freeze)
        I = train_interp.Interp(P, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, 1)
        logits = I.forward(torch.from_numpy(batch["img"]))
        for k, v in logits.items():
            assert float((v - ref["logits"][k]).abs().max()) < 1e-9
        lg = {k: v.clone().requires_grad_(True) for k, v in logits.items()}
        total, _ = train_torch.loss_terms(lg, {k: torch.as_tensor(v) for k, v in batch.items()}, nt, torch.float64)
        total.backward()
        grads = I.backward({k: v.grad for k, v in lg.items()})
        assert set(grads) == set(k for k, v in ref["grads"].items() if v is not None) == P.trainable
        for k, g in grads.items():
            r = ref["grads"][k]
            assert float((g - r).abs().max()) <= 1e-8 * (float(r.abs().max()) + 1e-30), k
        for k, v in ref["new_stats"].items():
            assert float((I.sd[k] - v).abs().max()) < 1e-9, k
    finally:
        torch.set_default_dtype(old)


def test_freeze_scoping_counts():
    # the reference's golden run: 262 of 407 parameters receive a gradient in phase 0 (freeze), all in phase 1
    assert len(TrainPlan("original", 5, True).trainable) == 262
    assert len(TrainPlan("original", 5, False).trainable) == 407


def test_plan_structure_invariants():
    """Dilated gradient buffers exist exactly behind stride-2 conv outputs; all running sums of a residual block share one
    gradient buffer; every gradient buffer is referenced by some backward op; layouts are disjoint and aligned."""
    from hover_net_amd import train_plan as TP
    P = TP.TrainPlan("original", 5, False)
    by_name = {b.name: b for b in P.bufs}
    for blk in ("d0", "d1", "d2", "d3"):
        sums = [b for b in P.bufs if b.name.startswith(blk + ".") and (b.name.endswith(".sum") or b.name == blk + ".shortcut")]
        assert len({id(b.g) for b in sums}) == 1 and len(sums) >= 4
        assert sums[0].gstep == (1 if blk == "d0" else 2)
    assert by_name["d1.units.0.z2"].gstep == 2 and by_name["d1.units.1.z2"].gstep == 1 and by_name["d0.units.0.z2"].gstep == 1
    used = set()
    for op in P.bwd:
        for v in (getattr(op, k, None) for k in ("dx", "dy", "dz", "da", "dlo", "dskip")):
            if v is not None:
                used.add(id(v.buf))
    assert used == {id(g) for g in P.gbufs}
    d, g = P.layout(3)
    offs = sorted((b.off, b.size(3)) for b in P.bufs)
    assert all(o % 64 == 0 for o, _ in offs) and all(o1 + s1 <= o2 for (o1, s1), (o2, _) in zip(offs, offs[1:])) and offs[-1][0] + offs[-1][1] <= d
    goffs = sorted((b.off, b.size(3)) for b in P.gbufs)
    assert all(o1 + s1 <= o2 for (o1, s1), (o2, _) in zip(goffs, goffs[1:])) and goffs[-1][0] + goffs[-1][1] <= g
    # phase 0: no encoder unit carries a gradient, d0's shortcut path does
    F = TP.TrainPlan("original", 5, True)
    fb = {b.name: b for b in F.bufs}
    assert fb["d0.units.1.z1"].g is None and fb["d1.out"].g is None and fb["d0.shortcut"].g is not None and fb["conv0.z"].g is not None


@pytest.mark.parametrize("mode,nt,freeze", [("original", 5, False), ("original", 5, True), ("fast", None, False)])
def test_first_writer_rule_is_sound(mode, nt, freeze):
    """train_plan._first_writers, checked statically for every configuration the engine builds: (a) an op marked `store` is the first op
    of the backward list that touches its destination buffer at all; (b) a buffer the engine does not clear (`zero` False) is written
    completely by such an op before any op reads it; (c) the layout puts the cleared buffers first, disjoint and aligned; (d) most of the
    arena needs no clearing (the point of the exercise).  test_lowering_matches_oracle_fp64 executes the same flags with NaN in the
    uncleared buffers against the autograd oracle."""
    P = TrainPlan(mode, nt, freeze)
    _, gelems = P.layout(2)
    seen_w, seen_r = {}, {}
    for i, op in enumerate(P.bwd):
        for k in ("dy", "da"):
            v = getattr(op, k, None)
            if v is not None:
                seen_r.setdefault(id(v.buf), i)
        for k in ("dx", "dz", "dlo", "dskip"):
            v = getattr(op, k, None)
            if v is None:
                continue
            if getattr(op, "store", False) and k in ("dx", "dz"):
                assert id(v.buf) not in seen_w and id(v.buf) not in seen_r, op.name
            seen_w.setdefault(id(v.buf), (i, op, v))
    n_store = 0
    for g in P.gbufs:
        if g.zero:
            continue
        i, op, v = seen_w[id(g)]
        assert op.store and op.kind in ("bnrelu_bwd", "dgrad"), g.name
        assert (v.y0, v.x0, v.c0, v.step, v.h, v.w, v.c) == (0, 0, 0, 1, g.h, g.w, g.c), g.name
        assert seen_r.get(id(g), 1 << 30) > i, g.name
        n_store += 1
    spans = sorted((g.off, g.off + g.size(2), g.zero) for g in P.gbufs)
    for (a0, a1, _), (b0, _, _) in zip(spans, spans[1:]):
        assert a1 <= b0 and b0 % 64 == 0
    assert all(z for o, _, z in spans if o < P.gzero_elems) and not any(z for o, _, z in spans if o >= P.gzero_elems)
    assert spans[-1][1] <= gelems and P.gzero_elems < 0.3 * gelems and n_store > 100


@pytest.mark.parametrize("mode,nt,freeze", [("original", 5, False), ("original", 5, True), ("fast", None, False)])
def test_weight_gradient_windows_hold_no_writer_of_their_output_gradient(mode, nt, freeze):
    """train_plan.wgrad_windows (what lets a weight gradient run on a second stream, train_engine._floating_wgrads): between a `wgrad`
    op and the end of its window no op writes the buffer its dy lives in, the op at the end of the window does, dy has a writer BEFORE the
    weight gradient (its dy is complete when the ops in front of it are)."""
    P = TrainPlan(mode, nt, freeze)
    win = P.wgrad_windows()
    assert len(win) == sum(1 for op in P.bwd if op.kind == "wgrad") > 50

    def written(op):
        return {id(getattr(op, k).buf) for k in ("dx", "dz", "dlo", "dskip") if getattr(op, k, None) is not None}
    open_ended = 0
    for i, j in win.items():
        g = id(P.bwd[i].dy.buf)
        end = len(P.bwd) if j is None else j
        assert all(g not in written(P.bwd[k]) for k in range(i + 1, end)), P.bwd[i].name
        if j is None:
            open_ended += 1
        else:
            assert j > i and g in written(P.bwd[j]), P.bwd[i].name
        assert any(g in written(P.bwd[k]) for k in range(i)), P.bwd[i].name            # its dy was produced by an earlier op of the list
    # a BN's grad z has one writer: the weight gradients of the convs in front of a BN float to the join; the ones reading a residual sum or a
    # dense block's concat end at the next accumulation into it
    assert open_ended > 30 and open_ended < len(win)


def test_section_runs_of_the_lowered_lists():
    """train_engine.TrainEngine._runs: maximal runs of equal section tags over lowered groups, in launch indices (empty groups -- an
    `upadd_bwd` whose sums are all deferred -- neither break nor start a run)."""
    from hover_net_amd.train_engine import TrainEngine
    runs = TrainEngine._runs([-1, -1, 0, 0, 0, 1, 1, -1, -1], [[1], [1, 2], [1], [], [1, 1], [1], [1, 1, 1], [], [1, 2, 3]])
    assert runs == [(-1, 0, 3), (0, 3, 6), (1, 6, 10), (-1, 10, 13)]
    assert TrainEngine._runs([], []) == [] and TrainEngine._runs([2], [[]]) == []

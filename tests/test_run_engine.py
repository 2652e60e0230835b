"""CPU: the run-loop protocol (hover_net_amd/run_engine.py = run_utils/engine.py:132-204 + the data-path callbacks of
run_utils/callbacks/base.py) wired the way opt.py:96-140 wires it, on fake step functions."""
import os

import torch

from hover_net_amd import run_engine as RE


class _Loader(list):
    batch_size = 2


def _wire(tmp_path=None):
    calls = []

    def step(batch, info):
        calls.append((batch, info[1]["epoch"], info[1]["step"]))
        assert set(info[0]["net"]) == {"desc", "optimizer", "lr_scheduler", "extra_info"}
        return {"EMA": {"overall_loss": float(batch)}, "raw": {"x": [batch, batch]}}

    net = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    sch = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.5)
    ri = {"net": {"desc": net, "optimizer": opt, "lr_scheduler": sch, "extra_info": {}}}
    tr = RE.RunEngine("train", _Loader([1, 2, 3]), step, ri)
    va = RE.RunEngine("valid", _Loader([10, 20]), step, ri)
    tr.add_event_handler(RE.Events.STEP_COMPLETED, RE.ScalarMovingAverage(alpha=0.5))
    tr.add_event_handler(RE.Events.EPOCH_COMPLETED, RE.TrackLr())
    if tmp_path is not None:
        tr.state.logging, tr.state.log_dir = True, str(tmp_path)
        tr.add_event_handler(RE.Events.EPOCH_COMPLETED, RE.PeriodicSaver())
    va.add_event_handler(RE.Events.STEP_COMPLETED, RE.AccumulateRawOutput())
    va.add_event_handler(RE.Events.EPOCH_COMPLETED, RE.ProcessAccumulatedRawOutput(lambda raw: {"scalar": {"n": len(raw["x"])}, "image": {}}))
    trig = RE.TriggerEngine("valid")
    trig.triggered_engine = va
    tr.add_event_handler(RE.Events.EPOCH_COMPLETED, trig)
    tr.add_event_handler(RE.Events.EPOCH_COMPLETED, RE.ScheduleLr())
    return tr, va, calls, opt


def test_engine_protocol_and_callback_wiring(tmp_path):
    tr, va, calls, opt = _wire(tmp_path)
    seen = []

    class Probe:                                    # any object with .run(state, event) is a handler (the reference's protocol)
        def run(self, state, event):
            seen.append((event, state.curr_epoch, state.curr_global_step))

    for ev in RE.Events:
        tr.add_event_handler(ev, Probe())
    tr.run(nr_epoch=2)
    # 2 epochs x (3 train steps + a chained validation run of 2 steps)
    assert [c[0] for c in calls] == [1, 2, 3, 10, 20, 1, 2, 3, 10, 20]
    assert [c[2] for c in calls if c[0] < 10] == [0, 1, 2, 3, 4, 5]            # global step keeps counting across epochs
    assert tr.state.curr_epoch == 2 and tr.state.curr_global_step == 6
    assert abs(opt.param_groups[0]["lr"] - 0.025) < 1e-12                      # ScheduleLr once per epoch
    ema = tr.state.tracked_step_output["scalar"]["overall_loss"]
    want = 1.0
    for v in (2, 3, 1, 2, 3):                                                  # the EMA runs on across epochs (callback state)
        want = 0.5 * want + 0.5 * v
    assert abs(ema - want) < 1e-12 and "lr-net" in tr.state.tracked_step_output["scalar"]
    assert va.state.tracked_step_output["scalar"] == {"n": 4}                  # 2 steps x 2 raw items, reset per chained run
    kinds = [e for e, _, _ in seen]
    assert kinds.count(RE.Events.EPOCH_STARTED) == 2 and kinds.count(RE.Events.STEP_STARTED) == 6 and kinds.count(RE.Events.EPOCH_COMPLETED) == 2
    for ep in (1, 2):                                                          # PeriodicSaver: reference checkpoint layout
        ck = torch.load(os.path.join(str(tmp_path), "net_epoch=%d.tar" % ep))
        assert sorted(ck) == ["desc", "lr_scheduler", "optimizer"] and "weight" in ck["desc"]

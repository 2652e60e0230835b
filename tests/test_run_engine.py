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


# ---- differential run against the reference's own engine + callbacks ---------------------------------------------------------------
_SCENARIO = r'''
import sys, types, json
impl, repo = sys.argv[1], sys.argv[2]
if impl == "ref":
    sys.path.insert(0, "/root/reference")
    for name in ("cv2", "termcolor"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["termcolor"].colored = lambda s, *a, **k: s
    from run_utils.engine import Events, RunEngine                       # the reference, unmodified
    from run_utils.callbacks.base import (AccumulateRawOutput, ProcessAccumulatedRawOutput, ScalarMovingAverage, ScheduleLr, TrackLr, TriggerEngine)
    import run_utils.engine as E
    assert E.__file__.startswith("/root/reference")
else:
    sys.path.insert(0, repo)
    from hover_net_amd.run_engine import (AccumulateRawOutput, Events, ProcessAccumulatedRawOutput, RunEngine, ScalarMovingAverage, ScheduleLr, TrackLr,
                                          TriggerEngine)
import torch


class Loader(list):
    batch_size = 2


trace = []


def train_step(batch, info):
    trace.append(["train_step", batch, info[1]["epoch"], info[1]["step"], sorted(info[0]["net"])])
    return {"EMA": {"overall_loss": batch * 0.5, "loss_np_bce": batch + 1.0}, "raw": {"x": [batch]}}


def valid_step(batch, info):
    trace.append(["valid_step", batch, info[1]["epoch"], info[1]["step"]])
    return {"raw": {"a": [batch, batch + 1], "b": [batch * 2]}}


class Probe:
    engine_trigger = False

    def __init__(self, tag):
        self.tag = tag

    def run(self, state, event):
        trace.append([self.tag, event.value, state.curr_epoch, state.curr_global_step, state.curr_epoch_step,
                      {k: float(v) for k, v in state.tracked_step_output["scalar"].items()},
                      {k: [float(x) for x in v] for k, v in state.epoch_accumulated_output.items()}, state.global_state is not None])


net = torch.nn.Linear(2, 2)
opt = torch.optim.SGD(net.parameters(), lr=0.1)
sch = torch.optim.lr_scheduler.StepLR(opt, 2, gamma=0.5)
ri = {"net": {"desc": net, "optimizer": opt, "lr_scheduler": sch, "extra_info": {}}}
tr = RunEngine(engine_name="train", dataloader=Loader([1, 2, 3]), run_step=train_step, run_info=ri)
va = RunEngine(engine_name="valid", dataloader=Loader([10, 20]), run_step=valid_step, run_info=ri)
tr.add_event_handler(Events.EPOCH_STARTED, Probe("train_start"))
tr.add_event_handler(Events.STEP_COMPLETED, ScalarMovingAverage(alpha=0.9))
tr.add_event_handler(Events.EPOCH_COMPLETED, TrackLr())
trig = TriggerEngine("valid")
assert trig.engine_trigger and trig.triggered_engine_name == "valid"
trig.triggered_engine = va                                             # run_train.py:250-257
tr.add_event_handler(Events.EPOCH_COMPLETED, trig)
tr.add_event_handler(Events.EPOCH_COMPLETED, Probe("train_epoch"))
tr.add_event_handler(Events.EPOCH_COMPLETED, ScheduleLr())
va.add_event_handler(Events.STEP_COMPLETED, AccumulateRawOutput())
va.add_event_handler(Events.EPOCH_COMPLETED, ProcessAccumulatedRawOutput(lambda acc: {"scalar": {"n": len(acc["a"]), "s": float(sum(acc["b"]))}, "image": {}}))
va.add_event_handler(Events.EPOCH_COMPLETED, Probe("valid_epoch"))
tr.run(3)
trace.append(["final", tr.state.curr_epoch, tr.state.curr_global_step, len(tr.state.run_accumulated_output), va.state.curr_epoch, va.state.curr_global_step,
              float(opt.param_groups[0]["lr"]), tr.state.batch_size])
print("TRACE " + json.dumps(trace))
'''


def test_engine_and_callbacks_equal_the_references_own_on_one_scenario():
    """run_utils/engine.py + run_utils/callbacks/base.py run unmodified (opt.py:96-140 wiring on fake steps) vs hover_net_amd.run_engine:
    the same sequence of step calls, event payloads, moving averages, learning rates, accumulated validation outputs and counters."""
    import json
    import subprocess
    import sys

    import pytest

    if not os.path.exists("/root/reference/run_utils/engine.py"):
        pytest.skip("needs the reference tree (build container only)")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    traces = {}
    for impl in ("ref", "mine"):
        r = subprocess.run([sys.executable, "-c", _SCENARIO, impl, repo], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg"))
        assert r.returncode == 0 and "TRACE " in r.stdout, (impl, r.stdout[-600:], r.stderr[-2500:])
        traces[impl] = json.loads(r.stdout.split("TRACE ", 1)[1].splitlines()[0])
    assert len(traces["ref"]) == len(traces["mine"]) > 20
    for a, b in zip(traces["ref"], traces["mine"]):
        assert a == b, (a, b)

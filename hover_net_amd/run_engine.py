"""The run-loop protocol of the reference (`run_utils/engine.py:132-204` RunEngine / Events / State and the data-path callbacks
of `run_utils/callbacks/base.py`): an engine walks a dataloader, calls `run_step(batch, [run_info, {"epoch", "step"}])`, and
fires handlers -- objects with `.run(state, event)` -- on six events.  `opt.py:96-140` wires training with exactly these
pieces: train engine {STEP_COMPLETED: ScalarMovingAverage; EPOCH_COMPLETED: TrackLr, PeriodicSaver, TriggerEngine("valid"),
ScheduleLr}, valid engine {STEP_COMPLETED: AccumulateRawOutput; EPOCH_COMPLETED: ProcessAccumulatedRawOutput}.

This module provides the same surface -- event names, `State` fields, `add_event_handler`, `run(nr_epoch, shared_state,
chained)`, the handler protocol -- so the reference's own callback objects (logging, visualisation) attach unchanged, and
the data-path callbacks listed above so `hover_net_amd.train.run_phases` is the reference's wiring on the HIP step functions.
Not rebuilt (host-only glue): tqdm progress bars, LoggingEpochOutput / LoggingGradient (tensorboard + json), VisualizeOutput,
ConditionalSaver."""
from enum import Enum

import torch


class Events(Enum):
    EPOCH_STARTED = "epoch_started"
    EPOCH_COMPLETED = "epoch_completed"
    STEP_STARTED = "step_started"
    STEP_COMPLETED = "step_completed"
    STARTED = "started"
    COMPLETED = "completed"
    EXCEPTION_RAISED = "exception_raised"


class State:
    """Shared between the handlers of one engine (engine.py:17-72: same field names)."""

    def __init__(self):
        self.logging = None
        self.log_dir = None
        self.log_info = None
        self.curr_epoch_step = 0
        self.curr_global_step = 0
        self.curr_epoch = 0
        self.tracked_step_output = {"scalar": {}, "image": {}}
        self.epoch_accumulated_output = {}
        self.run_accumulated_output = []
        self.step_output = None
        self.global_state = None
        self.pertain_n_epoch_output = 1

    def reset_variable(self):
        self.tracked_step_output = {k: {} for k in self.tracked_step_output}
        if self.curr_epoch % self.pertain_n_epoch_output == 0:
            self.run_accumulated_output = []
        self.epoch_accumulated_output = {}
        self.step_output = None


class RunEngine:
    def __init__(self, engine_name=None, dataloader=None, run_step=None, run_info=None, log_info=None):
        self.engine_name, self.run_step, self.dataloader = engine_name, run_step, dataloader
        self.state = State()
        self.state.attached_engine_name = engine_name
        self.state.run_info = run_info
        self.state.log_info = log_info
        self.state.batch_size = getattr(dataloader, "batch_size", None)
        self.event_handler_dict = {event: [] for event in Events}
        self.terminate = False

    def add_event_handler(self, event_name, handler):
        self.event_handler_dict[event_name].append(handler)

    def _trigger(self, event):
        for handler in self.event_handler_dict[event]:
            handler.run(self.state, event)

    def run(self, nr_epoch=1, shared_state=None, chained=False):
        if chained:                              # a triggered engine (validation) starts over every time
            self.state.curr_epoch = 0
        self.state.global_state = shared_state
        while self.state.curr_epoch < nr_epoch:
            self.state.reset_variable()
            self._trigger(Events.EPOCH_STARTED)
            for data_batch in self.dataloader:
                self._trigger(Events.STEP_STARTED)
                info = [self.state.run_info, {"epoch": self.state.curr_epoch, "step": self.state.curr_global_step}]
                self.state.step_output = self.run_step(data_batch, info)
                self._trigger(Events.STEP_COMPLETED)
                self.state.curr_global_step += 1
                self.state.curr_epoch_step += 1
            self.state.curr_epoch += 1
            self._trigger(Events.EPOCH_COMPLETED)
            self.state.run_accumulated_output.append(self.state.epoch_accumulated_output)


# ---- the data-path callbacks of run_utils/callbacks/base.py ----------------------------------------------------------------------
class BaseCallbacks:
    def __init__(self):
        self.engine_trigger = False

    def reset(self):
        pass

    def run(self, state, event):
        pass


class TrackLr(BaseCallbacks):
    def run(self, state, event):
        for net_name, net_info in state.run_info.items():
            state.tracked_step_output["scalar"]["lr-%s" % net_name] = net_info["optimizer"].param_groups[0]["lr"]


class ScheduleLr(BaseCallbacks):
    def run(self, state, event):
        for net_info in state.run_info.values():
            net_info["lr_scheduler"].step()


class TriggerEngine(BaseCallbacks):
    def __init__(self, triggered_engine_name, nr_epoch=1):
        super().__init__()
        self.engine_trigger = True
        self.triggered_engine_name = triggered_engine_name
        self.triggered_engine = None            # bound by the wiring code (run_train.py:250-254)
        self.nr_epoch = nr_epoch

    def run(self, state, event):
        self.triggered_engine.run(chained=True, nr_epoch=self.nr_epoch, shared_state=state)


class ScalarMovingAverage(BaseCallbacks):
    def __init__(self, alpha=0.95):
        super().__init__()
        self.alpha = alpha
        self.tracking_dict = {}

    def run(self, state, event):
        for key, value in state.step_output["EMA"].items():
            old = self.tracking_dict.get(key)
            self.tracking_dict[key] = value if old is None else old * self.alpha + (1.0 - self.alpha) * value
        state.tracked_step_output["scalar"] = self.tracking_dict


class AccumulateRawOutput(BaseCallbacks):
    def run(self, state, event):
        acc = state.epoch_accumulated_output
        for key, value in state.step_output["raw"].items():
            acc.setdefault(key, []).extend(list(value))


class ProcessAccumulatedRawOutput(BaseCallbacks):
    def __init__(self, proc_func, per_n_epoch=1):
        super().__init__()
        self.per_n_epoch = per_n_epoch
        self.proc_func = proc_func

    def run(self, state, event):
        state.tracked_step_output = self.proc_func(state.epoch_accumulated_output)


class PeriodicSaver(BaseCallbacks):
    """`{net_name}_epoch={n}.tar` = {key: value.state_dict()} for every entry of the net's run_info except extra_info
    (desc, optimizer, lr_scheduler), written by the process whose `state.logging` is set (rank 0)."""

    def __init__(self, per_n_epoch=1, per_n_step=None):
        super().__init__()
        self.per_n_epoch = per_n_epoch
        self.per_n_step = per_n_step

    def run(self, state, event):
        if not state.logging or state.curr_epoch % self.per_n_epoch != 0:
            return
        for net_name, net_info in state.run_info.items():
            ckpt = {}
            for key, value in net_info.items():
                if key == "extra_info":
                    continue
                sd = value.state_dict()
                if key == "desc":
                    sd = {k: v.detach().cpu().contiguous() for k, v in sd.items()}
                ckpt[key] = sd
            torch.save(ckpt, "%s/%s_epoch=%d.tar" % (state.log_dir, net_name, state.curr_epoch))

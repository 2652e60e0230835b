"""CPU (gloo, world_size 2): the collective wiring of run_desc.train_step -- SUM all-reduce of the loss partial
sums before the logit gradients and of the flat gradient slab after the backward pass -- with the GPU engine
replaced by a stub that records what it is handed.  The numerics of the two-rank step are covered on the GPU by
tests/test_gpu_train.py::test_two_rank_step_equals_dataparallel_semantics."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _StubEngine:
    def __init__(self, rank):
        self.rank = rank
        self.device = torch.device("cpu")
        self.sums = torch.zeros(64, dtype=torch.float64)
        self.gslab = torch.zeros(1000)
        self.logits = {"np": torch.zeros(2, 2, 4, 4), "hv": torch.zeros(2, 2, 4, 4)}
        self.world_seen = None

    def load_batch(self, batch):
        pass

    def forward(self):
        return self.logits

    def loss_and_backward(self, world=1, all_reduce=None):
        self.sums[:] = float(self.rank + 1)
        all_reduce(self.sums, False)
        self.world_seen = world
        self.gslab[:] = float(10 * (self.rank + 1))
        works = [all_reduce(self.gslab[600:], True), all_reduce(self.gslab[:600], True)]      # two buckets, async like the engine
        for w in works:
            w.wait()

    def loss_terms(self):
        return {"overall_loss": float(self.sums[0])}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hover_net_amd import run_desc, train_engine

    stub = _StubEngine(rank)
    train_engine.engine_for = lambda net, n: stub

    class Net(torch.nn.Module):
        nr_types, mode, freeze = None, "original", True

        def engine(self):  # marks the module as the HIP module for run_desc._unwrap
            return None

    class Opt:
        steps = 0

        def step(self):
            Opt.steps += 1

    net = Net()
    batch = {"img": torch.zeros(2, 8, 8, 3, dtype=torch.uint8), "np_map": torch.zeros(2, 4, 4), "hv_map": torch.zeros(2, 4, 4, 2)}
    out = run_desc.train_step(batch, [{"net": {"desc": net, "optimizer": Opt(), "extra_info": {"loss": {"np": {"bce": 1, "dice": 1}, "hv": {"mse": 1, "msge": 1}}}}}, {}])
    q.put((rank, float(stub.sums[0]), float(stub.gslab[0]), stub.world_seen, Opt.steps, out["EMA"]["overall_loss"], sorted(out["raw"].keys())))
    dist.destroy_process_group()


def test_train_step_all_reduces_sums_and_gradient_slab():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    for rank, s0, g0, world, steps, loss, raw_keys in res:
        assert s0 == 3.0 and g0 == 30.0           # 1 + 2, 10 + 20: SUM over both ranks
        assert world == 2 and steps == 1 and loss == 3.0
        assert raw_keys == ["hv", "img", "np"]


def test_unsupported_loss_weights_are_rejected():
    from hover_net_amd import run_desc

    class Net(torch.nn.Module):
        nr_types, mode, freeze = None, "original", True

        def engine(self):
            return None

    with pytest.raises(NotImplementedError):
        run_desc.train_step({"img": torch.zeros(1, 8, 8, 3)}, [{"net": {"desc": Net(), "optimizer": None, "extra_info": {"loss": {"np": {"bce": 2, "dice": 1}, "hv": {"mse": 1, "msge": 1}}}}}, {}])

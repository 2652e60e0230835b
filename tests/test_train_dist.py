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

    def set_loss_weights(self, loss_opts):
        self.loss_opts = loss_opts

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


def test_loss_weight_table_reaches_the_engine():
    """run_desc.py:66-82: `loss += weight * term` for the terms listed per branch.  train_step hands the table (restricted to
    the branches the network has) to the engine; the engine maps it onto the six weight slots of the C ABI (absent = 0) and
    rejects a term the reference's loss_func_dict does not have."""
    from hover_net_amd import lib as L
    from hover_net_amd import run_desc, train_engine

    stub = _StubEngine(0)
    orig = train_engine.engine_for
    train_engine.engine_for = lambda net, n: stub
    try:
        class Net(torch.nn.Module):
            nr_types, mode, freeze = None, "original", True

            def engine(self):
                return None

        class Opt:
            def step(self):
                pass

        stub.loss_and_backward = lambda world=1, all_reduce=None: None
        table = {"np": {"bce": 2, "dice": 1}, "hv": {"msge": 0.5}, "tp": {"bce": 1, "dice": 1}}
        batch = {"img": torch.zeros(2, 8, 8, 3, dtype=torch.uint8), "np_map": torch.zeros(2, 4, 4), "hv_map": torch.zeros(2, 4, 4, 2)}
        run_desc.train_step(batch, [{"net": {"desc": Net(), "optimizer": Opt(), "extra_info": {"loss": table}}}, {}])
        assert stub.loss_opts == {"np": {"bce": 2, "dice": 1}, "hv": {"msge": 0.5}}          # no tp branch in this network
    finally:
        train_engine.engine_for = orig

    class Fake:
        _WEIGHT_SLOT = train_engine.TrainEngine._WEIGHT_SLOT
        _loss = L.hvn_loss()

    f = Fake()
    train_engine.TrainEngine.set_loss_weights(f, {"np": {"bce": 2, "dice": 1}, "hv": {"msge": 0.5}})
    assert list(f._loss.weight) == [2.0, 1.0, 0.0, 0.5, 0.0, 0.0]
    train_engine.TrainEngine.set_loss_weights(f, None)
    assert list(f._loss.weight) == [1.0] * 6
    with pytest.raises(KeyError):
        train_engine.TrainEngine.set_loss_weights(f, {"np": {"focal": 1}})


def _bcast_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hover_net_amd import net_desc, train

    net = net_desc.create_model(mode="fast", nr_types=None, input_ch=3)       # every process draws its own random init
    before = float(sum(p.double().sum() for p in net.parameters()))
    train.broadcast_module_state(net)
    after = float(sum(p.double().sum() for p in net.parameters())) + float(sum(b.double().sum() for b in net.buffers()))
    ok_len = True
    try:
        train._same_on_all_ranks(7, "steps")
        train._same_on_all_ranks(7 + rank, "steps")
        ok_len = False
    except ValueError:
        pass
    q.put((rank, before, after, ok_len))
    dist.destroy_process_group()


def test_phase_start_broadcast_makes_ranks_identical():
    """each torchrun process draws its own Kaiming init; run_phases broadcasts rank 0's parameters and
    buffers before the engine binds them, and refuses loaders of different length across ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 400)
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    (_, b0, a0, ok0), (_, b1, a1, ok1) = res
    assert b0 != b1                      # different random inits ...
    assert a0 == a1                      # ... identical after the broadcast
    assert ok0 and ok1


def test_checkpoint_prefix_and_pretrained_config():
    from hover_net_amd import train

    sd = {"module.a": 1, "module.b.c": 2}
    assert train.convert_checkpoint_keys(sd) == {"a": 1, "b.c": 2}
    assert train.convert_checkpoint_keys({"a": 1, "module.b": 2}) == {"a": 1, "module.b": 2}     # only when every key has it
    cfg = train.get_config(5, "original", pretrained="/some/ImageNet-ResNet50-Preact_pytorch.tar")
    assert cfg["phase_list"][0]["run_info"]["net"]["pretrained"].endswith("Preact_pytorch.tar")
    assert cfg["phase_list"][1]["run_info"]["net"]["pretrained"] == -1
    with pytest.raises(ValueError, match="frozen random encoder"):
        train.run_phases(train.get_config(None, "fast"), lambda pi, bs: {"train": [], "valid": None}, device="cpu")

"""GPU: the multi-rank tile and whole-slide paths with the REAL kernels -- two processes (one rank each) share the box's single GPU and talk
over gloo, so everything but the transport is what an 8-GPU RCCL run executes: row-slab ownership of the prediction map, the halo
exchange, tile ownership by rows, per-patch maps routed to the stitching owner, the fan-in to rank 0, the on-device merge with uploaded
remote tiles.  Rank 0's results must equal the single-process run bit for bit.  (RCCL itself cannot run here: one GPU per box.)"""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _net():
    from hover_net_amd import net_desc
    from hover_net_amd.synth import synth_state_dict

    net = net_desc.create_model(mode="original", nr_types=5, input_ch=3)
    net.load_state_dict(synth_state_dict("original", 5, seed=81), strict=True)
    sd = net.state_dict()
    sd["decoder.np.u0.conv.bias"] = torch.tensor([-0.4, 0.4])       # random-init output near the nucleus threshold: blobs to separate
    net.load_state_dict(sd, strict=True)
    return net.to("cuda").eval()


def _slide():
    from hover_net_amd import infer_wsi

    rng = np.random.default_rng(82)
    return infer_wsi.ArraySlide(rng.integers(0, 256, (1150, 1010, 3), dtype=np.uint8))


def _images():
    rng = np.random.default_rng(7)
    return [rng.integers(0, 256, s, dtype=np.uint8) for s in ((300, 283, 3), (270, 270, 3), (401, 333, 3))]


def _work():
    from hover_net_amd import infer_tile, infer_wsi

    net = _net()
    wsi = infer_wsi.WsiInference(net, nr_types=5, batch_size=16, chunk_shape=700, tile_shape=512, ambiguous_size=64)
    inst, info = wsi.run(_slide(), mask=None)
    tiles = infer_tile.process_images(_images(), net, nr_types=5, batch_size=8)
    return (inst, None if info is None else sorted(info), wsi.map_rows_resident,
            [None if t is None else (t[0], sorted(t[1])) for t in tiles])


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _work()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    inst1, keys1, rows1, tiles1 = _work()                           # the same work in this process, one rank
    inst0, keys0, rows0, tiles0 = out[0]
    assert out[1][0] is None and out[1][1] is None                  # results live on rank 0
    assert np.array_equal(inst0, inst1) and keys0 == keys1 and len(keys1) > 20
    assert rows0 < inst1.shape[0] and out[1][2] < inst1.shape[0] and rows1 == inst1.shape[0]      # each rank held slab + halo only
    for a, b in zip(tiles0, tiles1):                                # tile path: every image complete on rank 0
        assert np.array_equal(a[0], b[0]) and a[1] == b[1]


@pytest.mark.parametrize("launch", ["torchrun", "self"])
def test_bench_step_on_two_ranks_sharing_the_gpu(launch):
    """The driver's exact N > 1 command path -- `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` -- with the REAL
    kernels and bench.py's OWN step (network, instance separation, the per-batch gather to rank 0 on the side stream inside the timed
    step, D2H), on this box's one GPU: HVN_BENCH_SHARED_GPU=1 puts both ranks on cuda:0 and the collectives on gloo (device tensors
    staged through the host, `infer_tile.gather_to_rank0`).  Nothing measured here is comparable; what is checked is that the line is
    produced, carries both ranks' tiles and the per-rank step / gather times a bad scaling curve would be diagnosed from."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HVN_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    port = 32000 + os.getpid() % 2000
    # launch = "torchrun": under the launcher, as the driver starts N > 1; "self": `python bench.py --gpus 2` alone -- the script becomes
    # its own launcher (the shape of the driver's N = 1 command).  The default (fitted) checkpoint path: rank 0 fits, the others receive it.
    args = [os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-variants",
            "--no-roofline", "--fit-steps", "12", "--wsi-leg", "--wsi-size", "4096"]
    cmd = [sys.executable] + (["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                               "--master-port", str(port)] if launch == "torchrun" else []) + args
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["steps"] == 3 and d["value"] > 0
    pr = d["config"]["per_rank"]
    assert len(pr["step_ms"]) == 2 and len(pr["gather_ms"]) == 2 and all(t > 0 for t in pr["step_ms"]) and all(t >= 0 for t in pr["gather_ms"])
    assert d["config"]["instances_last_step"] > 0          # rank 0 holds the gathered results of both ranks
    assert d["config"]["checkpoint"]["kind"].startswith("fitted") and "rank 0 and broadcast" in d["config"]["checkpoint"]["kind"]
    # round 6: BASELINE cfg 4 as a multi-rank leg -- row-slab ownership, ONE halo all_to_all, owner-post-processed tiles, rank 0's merge
    w = d["variants"]["wsi_4k"]
    assert w["world_size"] == 2 and len(w["per_rank"]) == 2 and w["patches"] == sum(p["patches"] for p in w["per_rank"]) > 2000 and w["instances"] > 100
    assert all(p["patches"] > 0 and p["stage1_s"] > 0 and p["stage2_s"] > 0 and p["halo_exchange_s"] > 0 for p in w["per_rank"])
    assert sum(p["halo_rows"] for p in w["per_rank"]) > 0                      # (the last slab's tiles reach into nobody's rows)
    assert all(p["map_rows_resident"] < 4096 for p in w["per_rank"])          # each rank held its slab + halo, not the map

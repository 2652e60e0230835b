"""CPU (gloo, 2 ranks): the distributed control flow of bench.py -- launched exactly as the driver launches it for N > 1
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`) with
`--control-flow-selftest`, which swaps the GPU pipeline for correctly shaped dummy results and RCCL for gloo: barriers, the
max-over-ranks clock, the per-batch `gather_to_rank0` inside the step (rank order checked), the sustained leg's stop vote and the
closing barrier must all line up on every rank, or this hangs / fails here instead of on the 8-GPU box."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_rank_control_flow():
    port = 31000 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--control-flow-selftest"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 prints ONE json line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    assert out["config"]["world_size"] == 2 and out["config"]["global_batch"] == 64 and out["config"]["sustained_steps"] >= 4


def test_bench_launches_itself_when_no_launcher_is_around():
    """`python bench.py --gpus 2` (the shape of the driver's N = 1 command with another N): no WORLD_SIZE in the environment, so the script
    starts itself under torch.distributed.run on a free port and the two ranks run the same control flow; ONE json line comes back."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--control-flow-selftest"],
                       capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["world_size"] == 2


def test_bench_refuses_a_world_size_mismatch():
    """Under a launcher whose world size is not --gpus the script stops (it does not silently run another job)."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(33000 + os.getpid() % 2000))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--control-flow-selftest"], capture_output=True, text=True,
                       timeout=300, cwd=REPO, env=env)
    assert r.returncode != 0 and "nproc-per-node 2" in r.stderr

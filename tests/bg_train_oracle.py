"""Test infrastructure: the training oracle's CPU runs (oracle/train_torch.py; the float64 one takes 40 - 70 s) computed by a background
process while the GPU tests that precede tests/test_gpu_train.py run, so that the suite's wall time is not the sum of the two.

`start()` is called when test_gpu_train.py is imported on a GPU box (collection time); `get(case, dtype)` returns the result dict, computing
it in-process if the worker is not there, failed, or does not finish.  The worker sees no GPU (HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
empty) and takes a bounded number of CPU threads.  Run as a script it is the worker:  python bg_train_oracle.py <out_dir> <threads> <case:dtype> ..."""
import os
import subprocess
import sys
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
_PROC = None
_DIR = None
# in the order test_gpu_train.py asks for them (round-5 advisor: the float64 run is made for EVERY case again -- it costs the suite nothing
# while it runs behind the inference tests that conftest.py orders in front of the training tests)
JOBS = (("orig5_freeze", "float32"), ("orig5_freeze", "float64"), ("orig5_full", "float32"), ("orig5_full", "float64"),
        ("fastseg_full", "float32"), ("fastseg_full", "float64"))


def _inputs(case):
    sys.path.insert(0, _HERE)
    from test_oracle_train import load_case
    from hover_net_amd.synth import synth_state_dict, synth_train_batch

    gold, mode, nt, freeze = load_case(case)
    sd = synth_state_dict(mode, nt, seed=int(gold["wseed"]))
    batch = synth_train_batch(int(gold["n"]), mode, nt, seed=int(gold["bseed"]))
    return sd, batch, mode, nt, freeze


def compute(case, dtype):
    import torch
    from oracle import train_torch

    sd, batch, mode, nt, freeze = _inputs(case)
    return train_torch.train_step(sd, batch, mode, nt, freeze, dtype=getattr(torch, dtype))


def start():
    """Spawn the worker (once per process); silently a no-op where that is not possible."""
    global _PROC, _DIR
    if _PROC is not None or os.environ.get("HVN_TRAIN_ORACLE_BG", "1") == "0":
        return
    try:
        _DIR = tempfile.mkdtemp(prefix="hvn_oracle_")
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", PYTHONPATH=os.pathsep.join([os.path.dirname(_HERE), _HERE, os.environ.get("PYTHONPATH", "")]))
        # (at most 16: torch's CPU convolutions do not scale further, and the GPU tests in the foreground run CPU oracles of their own)
        threads = min(16, max(4, (os.cpu_count() or 8) // 2))
        _PROC = subprocess.Popen([sys.executable, os.path.abspath(__file__), _DIR, str(threads)] + ["%s:%s" % j for j in JOBS], env=env,
                                 stdout=subprocess.DEVNULL, stderr=open(os.path.join(_DIR, "worker.err"), "w"))
        import atexit

        atexit.register(_stop)                         # a run that ends early (-x) must not leave the worker computing behind whatever runs next
    except Exception:                                  # noqa: BLE001 -- the in-process path covers every failure
        _PROC = None


def _stop():
    import shutil

    if _PROC is not None and _PROC.poll() is None:
        _PROC.terminate()
        try:
            _PROC.wait(timeout=10)
        except Exception:                              # noqa: BLE001
            _PROC.kill()
    if _DIR:
        shutil.rmtree(_DIR, ignore_errors=True)        # (the float64 result alone is ~300 MB)


def get(case, dtype, timeout=600):
    import torch

    path = os.path.join(_DIR, "%s_%s.pt" % (case, dtype)) if _DIR else None
    if _PROC is not None and path:
        import time

        t0 = time.time()
        while not os.path.exists(path) and _PROC.poll() is None and time.time() - t0 < timeout:
            time.sleep(0.5)
        if os.path.exists(path):
            try:
                return torch.load(path, weights_only=False)
            except Exception:                          # noqa: BLE001
                pass
    return compute(case, dtype)


if __name__ == "__main__":
    out_dir, threads = sys.argv[1], int(sys.argv[2])
    import torch

    torch.set_num_threads(threads)
    for job in sys.argv[3:]:
        case, dtype = job.split(":")
        res = compute(case, dtype)
        res = {k: ({kk: (vv.detach() if hasattr(vv, "detach") else vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in res.items()}
        tmp = os.path.join(out_dir, "%s_%s.pt.tmp" % (case, dtype))
        torch.save(res, tmp)
        os.replace(tmp, os.path.join(out_dir, "%s_%s.pt" % (case, dtype)))

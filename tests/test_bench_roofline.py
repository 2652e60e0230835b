"""CPU: bench.py's roofline accounting (`roofline_account`) on made-up launch times -- the arithmetic behind the `roofline` object of the
driver's JSON line, for the mixed-pipe fp32 engine (bf16x3 launches on the bf16 matrix pipe, the rest on the fp32 pipe, Winograd transforms
on neither) and for the one-pipe cases (HVN_X3=0)."""
import importlib.util
import os

import numpy as np

from hover_net_amd import plan as PL
from hover_net_amd.synth import synth_state_dict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _timed(P):
    return [o for o in P.ops if o.kind in (PL.OP_CONV, PL.OP_CHAIN, PL.OP_WINO_IN, PL.OP_WINO_OUT)]


def test_mixed_pipe_accounting_recovers_the_rates_it_was_fed():
    P = PL.build_plan(synth_state_dict("original", 5, seed=0), "original", 5)
    timed, batch = _timed(P), 32
    # made-up machine: bf16x3 conv launches run at 1000 TFLOP/s of bf16 MFMA, the bf16x3 chained seams (round 5) at 500, fp32-pipe launches
    # at 78.65 (half the peak), every transform 0.05 ms
    per = []
    for o in timed:
        fl = o.extra.get("exec_flops", o.flops()) * batch if o.kind in (PL.OP_CONV, PL.OP_CHAIN) else 0.0
        if o.kind == PL.OP_CONV and o.extra.get("x3"):
            per.append(fl * o.extra["x3"] / 1000e12 * 1e3)
        elif o.kind == PL.OP_CHAIN and o.extra.get("x3"):
            per.append(fl * o.extra["x3"] / 500e12 * 1e3)
        elif fl:
            per.append(fl / 78.65e12 * 1e3)
        else:
            per.append(0.05)
    r = bench.roofline_account(timed, np.array(per), batch, "fp32")
    n_tr = sum(1 for o in timed if o.kind in (PL.OP_WINO_IN, PL.OP_WINO_OUT))
    assert r["launches"] == 93 and r["conv_launches_per_step"] == 133 and r["timed_launches_per_step"] == len(timed)
    assert abs(r["achieved"] - 1000.0) < 1e-6 and abs(r["frac"] - 0.4) < 1e-9 and r["peak"] == 2500.0
    assert abs(r["fp32_equivalent_tflops"] - 1000.0 / 6) < 1e-6
    o = r["other_launches"]
    t_fp32 = o["executed_gflop_per_step"] * 1e9 / 78.65e12 * 1e3
    assert abs(o["ms_per_step"] - (t_fp32 + 0.05 * n_tr)) < 1e-9 and abs(o["achieved"] - o["executed_gflop_per_step"] / o["ms_per_step"]) < 1e-6
    ch = r["chained_seams"]
    assert ch["launches"] == 3 and abs(ch["achieved"] - 500.0) < 1e-6 and abs(ch["frac"] - 0.2) < 1e-9 and abs(ch["fp32_equivalent_tflops"] - 500.0 / 6) < 1e-6
    w = r["whole_step"]
    assert abs(w["conv_ms_per_step"] - sum(per)) < 1e-9 and abs(r["ms_per_step"] + ch["ms_per_step"] + o["ms_per_step"] - sum(per)) < 1e-9
    ideal = 0.4 * r["ms_per_step"] + 0.2 * ch["ms_per_step"] + 0.5 * t_fp32
    assert abs(w["ideal_matrix_ms"] - ideal) < 1e-9 and abs(w["frac"] - ideal / sum(per)) < 1e-12
    assert abs(r["executed_gflop_per_step"] - (r["fp32_products_gflop_per_step"] + ch["bf16_mfma_gflop_per_step"] / 6 + o["executed_gflop_per_step"])) < 1e-6
    assert abs(r["algorithmic_gflop_per_step"] / batch - (392.17 - 1.31 - 0.006)) < 0.01           # SURVEY 8d's figure minus conv0 and the heads
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):                                # the bench contract's keys
        assert k in r


def test_one_pipe_accounting_without_bf16x3(monkeypatch):
    monkeypatch.setenv("HVN_X3", "0")
    P = PL.build_plan(synth_state_dict("original", 5, seed=0), "original", 5)
    timed, batch = _timed(P), 32
    per = np.full(len(timed), 0.3)
    r = bench.roofline_account(timed, per, batch, "fp32")
    assert r["peak"] == 157.3 and "other_launches" not in r and abs(r["conv_ms_per_step"] - 0.3 * len(timed)) < 1e-9
    assert abs(r["achieved"] - r["executed_gflop_per_step"] / r["conv_ms_per_step"]) < 1e-6 and abs(r["frac"] - r["achieved"] / 157.3) < 1e-12

#!/usr/bin/env python
"""Roofline fraction of the training step's MFMA kernels from a rocprofv3 --kernel-trace results.db of
`python tools/train_bench.py --phase P --steps K --warmup W` and the JSON line that run printed:
    frac = executed MFMA FLOPs per step (train_bench's count) / (time of the MFMA kernels per step) / 157.3 TFLOP/s
MFMA kernels = hvn_conv_igemm_f32 (forward convs, data gradients, Winograd-domain products), hvn_conv_wgrad_f32, hvn_conv0_mfma,
hvn_conv0_wgrad_mfma, hvn_dense_grouped*; the Winograd transform launches are counted into the time (they exist only to feed them).
usage: python tools/train_roofline.py <results.db> <train_bench.jsonl> <steps + warmup of that run>"""
import json
import sqlite3
import sys

db, line, passes = sys.argv[1], json.loads(open(sys.argv[2]).readline()), int(sys.argv[3])
c = sqlite3.connect(db)
# conv launches before the first step (= before the first conv0 kernel) and weight-gradient launches are TrainEngine.autotune_tiles' timing launches: left out
first = c.execute("select min(start) from kernels where name like '%hvn_conv0%'").fetchone()[0] or 0
rows = list(c.execute("select name, count(*), sum(duration) from kernels where not (start < ? and (name like '%hvn_conv_igemm%' or name like '%hvn_conv_wgrad%')) group by name", (first,)))
mfma = ("hvn_conv_igemm", "hvn_conv_wgrad", "hvn_conv0_mfma", "hvn_conv0_wgrad_mfma", "hvn_dense_grouped", "hvn_conv_chain")
feed = ("hvn_wino_in", "hvn_wino_out", "hvn_wino_dy", "hvn_wino_dw", "hvn_pack_w")
t_mfma = sum(r[2] for r in rows if any(k in r[0] for k in mfma)) / 1e6 / passes
t_feed = sum(r[2] for r in rows if any(k in r[0] for k in feed)) / 1e6 / passes
t_all = sum(r[2] for r in rows) / 1e6 / passes
t_x3 = sum(r[2] for r in rows if "_x3" in r[0] and any(k in r[0] for k in mfma)) / 1e6 / passes      # launches whose products run on the bf16 pipe (bf16x3)
t_bn = sum(r[2] for r in rows if "hvn_bn_" in r[0]) / 1e6 / passes
n_bn = sum(r[1] for r in rows if "hvn_bn_" in r[0]) / passes
ex = line["executed_gflop_forward"] + line["executed_gflop_backward"]
print(json.dumps({"phase": line["phase"], "batch": line["batch"], "kernel_ms_per_step": t_all, "mfma_kernel_ms_per_step": t_mfma,
                  "winograd_transform_and_pack_ms_per_step": t_feed, "executed_gflop_per_step": ex,
                  "fp32_equivalent_tflops": ex / (t_mfma + t_feed), "fp32_equivalent_tflops_mfma_kernels_only": ex / t_mfma,
                  "bf16x3_kernel_ms_per_step": t_x3, "batchnorm_ms_per_step": t_bn, "batchnorm_launches_per_step": n_bn,
                  "conv_kernel_share_of_kernel_time": (t_mfma + t_feed) / t_all,
                  "note": "executed fp32 multiply-adds of the step's GEMMs (each product once) / (MFMA kernels + Winograd transforms + weight packing); since "
                          "round 4 the forward / data-gradient convs and since round 5 the weight gradients form their products on the bf16 pipe (6 bf16 MFMAs "
                          "per product), so this is NOT a fraction of one pipe's peak (fp32 matrix peak 157.3, bf16 2500 TFLOP/s); rocprofv3 --kernel-trace durations"}))

#!/bin/bash
# last refresh of the round: training bench and the cfg-3 lines on the final conv kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 70 python tools/train_bench.py --steps 8 --warmup 3 > gpurun_out/g29_train_bench.jsonl 2> gpurun_out/g29_train_bench.err
timeout 40 python bench.py --dtype bf16 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/g29_bench_cfg3_bf16.log 2>&1
timeout 40 python bench.py --dtype fp32 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/g29_bench_cfg3_fp32.log 2>&1
cut -c1-220 gpurun_out/g29_train_bench.jsonl; tail -1 gpurun_out/g29_bench_cfg3_bf16.log | cut -c1-200; tail -1 gpurun_out/g29_bench_cfg3_fp32.log | cut -c1-200

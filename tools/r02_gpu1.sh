#!/bin/bash
# round-2 GPU call 1: new parity tests + the new bench line + single-stream kernel trace (baseline for the kernel work)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_net.py tests/test_gpu_bench_shapes.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/g1_tests.log
timeout 600 python bench.py > gpurun_out/g1_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/g1_prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/g1_prof.log 2>&1
python tools/kernel_stats.py gpurun_out/g1_prof/r_results.db "bench.py --steps 3 --warmup 1 (single stream)" > gpurun_out/g1_kernel_stats.csv 2>gpurun_out/g1_ks.err
python tools/layer_table.py gpurun_out/g1_prof/r_results.db 32 > gpurun_out/g1_layer_table.txt 2>gpurun_out/g1_lt.err
rm -rf gpurun_out/g1_prof
cat gpurun_out/g1_tests.log; tail -c 3000 gpurun_out/g1_bench.log; tail -5 gpurun_out/g1_layer_table.txt

#!/bin/bash
# round-2 GPU call 10: big-window wave replay -- exactness, timing, cfg-3 lines, train bench, default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_bf16.py::test_bf16_sized_map_perturbation_keeps_the_segmentation tests/test_gpu_net.py::test_wsi_pipeline_on_synthetic_slide -q -m gpu -x 2>&1 | tail -6 > gpurun_out/g10_tests.log
for w in 1 0; do for a in "32 80 2 8" "32 80 5 40" "2 1000 2 8" "64 164 2 8"; do echo "wave=$w" >> gpurun_out/g10_pp.log; HVN_WS_WAVE=$w timeout 120 python tools/pp_bench.py $a 2>&1 | grep separate >> gpurun_out/g10_pp.log; done; done
timeout 300 python bench.py --dtype bf16 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g10_bench_cfg3_bf16.log 2>&1
timeout 300 python bench.py --dtype fp32 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g10_bench_cfg3_fp32.log 2>&1
timeout 300 python tools/train_bench.py --steps 8 --warmup 3 > gpurun_out/g10_train_bench.jsonl 2> gpurun_out/g10_train_bench.err
timeout 600 python tools/wsi_bench.py --size 40000 --skip-stage1 > gpurun_out/g10_wsi40k.log 2>&1
timeout 600 python bench.py > gpurun_out/g10_bench.log 2>&1
cat gpurun_out/g10_tests.log; paste - - < gpurun_out/g10_pp.log
for f in g10_bench g10_bench_cfg3_bf16 g10_bench_cfg3_fp32; do tail -1 gpurun_out/$f.log | cut -c1-220; done
cut -c1-160 gpurun_out/g10_train_bench.jsonl; tail -1 gpurun_out/g10_wsi40k.log | cut -c1-700

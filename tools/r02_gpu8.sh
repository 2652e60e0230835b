#!/bin/bash
# round-2 GPU call 8: wave replay with per-component tie fallback
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_bf16.py::test_bf16_sized_map_perturbation_keeps_the_segmentation -q -m gpu -x -s 2>&1 | tail -6 > gpurun_out/g8_tests.log
for w in 1 0; do for a in "32 80 2 8" "32 80 5 40" "2 1000 2 8" "64 164 2 8"; do echo "wave=$w" >> gpurun_out/g8_pp.log; HVN_WS_WAVE=$w timeout 120 python tools/pp_bench.py $a 2>&1 | grep separate >> gpurun_out/g8_pp.log; done; done
timeout 300 python tools/wsi_bench.py --size 8192 --skip-stage1 > gpurun_out/g8_wsi8k.log 2>&1
timeout 600 python tools/wsi_bench.py --size 40000 --skip-stage1 > gpurun_out/g8_wsi40k.log 2>&1
cat gpurun_out/g8_tests.log; paste - - < gpurun_out/g8_pp.log; tail -1 gpurun_out/g8_wsi8k.log; tail -1 gpurun_out/g8_wsi40k.log

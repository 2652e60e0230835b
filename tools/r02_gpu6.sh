#!/bin/bash
# round-2 GPU call 6: two-class watershed launches -- exactness, post-proc timing at three sizes, WSI stage 2 at 8192^2 and 40 000^2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_bf16.py::test_bf16_sized_map_perturbation_keeps_the_segmentation -q -m gpu -x 2>&1 | tail -6 > gpurun_out/g6_tests.log
for a in "32 80 2 8" "32 80 5 40" "2 1000 2 8" "1 2048 2 6"; do timeout 120 python tools/pp_bench.py $a >> gpurun_out/g6_pp.log 2>&1; done
timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/g6_prof -o r -- python tools/pp_bench.py 32 80 2 8 > /dev/null 2>&1
python tools/kernel_stats.py gpurun_out/g6_prof/r_results.db "pp_bench.py 32 80 2 8" > gpurun_out/g6_pp_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/g6_prof
timeout 300 python tools/wsi_bench.py --size 8192 --skip-stage1 > gpurun_out/g6_wsi8k.log 2>&1
timeout 600 python tools/wsi_bench.py --size 40000 --skip-stage1 > gpurun_out/g6_wsi40k.log 2>&1
cat gpurun_out/g6_tests.log; grep separate gpurun_out/g6_pp.log; head -8 gpurun_out/g6_pp_kernel_stats.csv; tail -1 gpurun_out/g6_wsi8k.log; tail -1 gpurun_out/g6_wsi40k.log

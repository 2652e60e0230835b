#!/bin/bash
# round-2 GPU call 7: wave-cooperative watershed replay -- exactness, A/B timing, WSI stage 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_bf16.py::test_bf16_sized_map_perturbation_keeps_the_segmentation tests/test_gpu_net.py::test_wsi_pipeline_on_synthetic_slide tests/test_gpu_net.py::test_two_stream_pipeline_equals_sequential tests/test_gpu_bench_shapes.py::test_pipeline_batch32_equals_sequential_with_host_output -q -m gpu -x -s 2>&1 | tail -12 > gpurun_out/g7_tests.log
for w in 1 0; do for a in "32 80 2 8" "32 80 5 40" "2 1000 2 8" "64 164 2 8"; do echo "wave=$w" >> gpurun_out/g7_pp.log; HVN_WS_WAVE=$w timeout 120 python tools/pp_bench.py $a 2>&1 | grep separate >> gpurun_out/g7_pp.log; done; done
timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/g7_prof -o r -- python tools/pp_bench.py 32 80 2 8 > /dev/null 2>&1
python tools/kernel_stats.py gpurun_out/g7_prof/r_results.db "pp_bench.py 32 80 2 8 (wave replay)" > gpurun_out/g7_pp_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/g7_prof
timeout 300 python tools/wsi_bench.py --size 8192 --skip-stage1 > gpurun_out/g7_wsi8k.log 2>&1
timeout 600 python tools/wsi_bench.py --size 40000 --skip-stage1 > gpurun_out/g7_wsi40k.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g7_bench.log 2>&1
cat gpurun_out/g7_tests.log; cat gpurun_out/g7_pp.log; head -6 gpurun_out/g7_pp_kernel_stats.csv | cut -c1-120; tail -1 gpurun_out/g7_wsi8k.log; tail -1 gpurun_out/g7_wsi40k.log; tail -1 gpurun_out/g7_bench.log | cut -c1-1800

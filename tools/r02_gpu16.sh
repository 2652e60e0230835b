#!/bin/bash
# round-2 GPU call 16: restructured conv epilogue -- parity, bench, layer table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_net.py tests/test_gpu_bench_shapes.py tests/test_gpu_train.py::test_training_step_matches_oracle -q -m gpu -x 2>&1 | tail -5 > gpurun_out/g16_tests.log
for shape in "32 66 256 1024 1 res" "32 264 64 256 1 res" "32 66 1024 256 1 pre" "32 66 256 256 3" "32 62 1024 256 5"; do
    HVN_TILE_SELECT=0 timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" >> gpurun_out/g16_conv.log
done
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g16_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/g16_prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/g16_prof.log 2>&1
python tools/kernel_stats.py gpurun_out/g16_prof/r_results.db "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants" > gpurun_out/g16_kernel_stats.csv 2>/dev/null
python tools/layer_table.py gpurun_out/g16_prof/r_results.db 32 > gpurun_out/g16_layer_table.txt 2>/dev/null
rm -rf gpurun_out/g16_prof
cat gpurun_out/g16_tests.log gpurun_out/g16_conv.log; tail -1 gpurun_out/g16_bench.log | cut -c1-1900; grep -E "^(d0|d1|d2|d3|conv_bot|decoder.tp.u3.(conva|dense)) " gpurun_out/g16_layer_table.txt

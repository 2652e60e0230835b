#!/bin/bash
# augmentation kernels (new) + a clean kernel-stats CSV (tile-autotune launches separated) + the bench line with the current PMC traffic file
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_augment.py tests/test_gpu_targets.py -q -x 2>&1 | tail -15 > gpurun_out/g24_tests.log
timeout 300 python -m pytest tests/test_gpu_train.py -q -x -k "valid or two_phase" 2>&1 | tail -3 >> gpurun_out/g24_tests.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/g24_prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/g24_prof.log 2>&1
python tools/kernel_stats.py gpurun_out/g24_prof/r_results.db "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants" > gpurun_out/g24_kernel_stats.csv 2>gpurun_out/g24_ks.err
python tools/layer_table.py gpurun_out/g24_prof/r_results.db 32 > gpurun_out/g24_layer_table.txt 2>/dev/null
rm -rf gpurun_out/g24_prof
cat gpurun_out/g24_tests.log; head -4 gpurun_out/g24_kernel_stats.csv | cut -c1-160; tail -1 gpurun_out/g24_prof.log | cut -c1-300

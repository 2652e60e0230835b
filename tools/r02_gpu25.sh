#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_augment.py tests/test_gpu_targets.py tests/test_gpu_net.py -q -k "augment or targets or loader or kernel or process_file_list or valid_step" --durations=5 2>&1 | tail -15 > gpurun_out/g25_tests.log
cat gpurun_out/g25_tests.log

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_augment.py tests/test_gpu_bench_shapes.py tests/test_gpu_bf16.py -q -s -m gpu 2>&1 | grep -E "passed|failed|segmentation|^E |Error" | head -30

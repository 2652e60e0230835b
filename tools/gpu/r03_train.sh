#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -k "matches_oracle" -s 2>&1 | grep -E "passed|failed|Winograd vs direct|Error|assert" | head -20



#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_two_ranks_one_gpu.py -q -x 2>&1 | grep -vE "^\[Gloo\]|amdgpu.ids" | tail -25

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 -p no:cacheprovider > gpurun_out/r06_tests_g.log 2>&1
tail -22 gpurun_out/r06_tests_g.log

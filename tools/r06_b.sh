#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/wgrad_static_rule.py > gpurun_out/r06_wgrad_static_rule.txt 2>&1
tail -5 gpurun_out/r06_wgrad_static_rule.txt
timeout 1500 python -m pytest tests -q -m gpu -x --durations=25 -p no:cacheprovider > gpurun_out/r06_tests_ordered.log 2>&1
tail -45 gpurun_out/r06_tests_ordered.log

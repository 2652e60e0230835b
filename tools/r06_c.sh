#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=20 -p no:cacheprovider > gpurun_out/r06_tests_f.log 2>&1
tail -32 gpurun_out/r06_tests_f.log
timeout 600 python tools/wgrad_terms_ab.py 2>/dev/null > gpurun_out/r06_wgrad_terms_ab.txt; cat gpurun_out/r06_wgrad_terms_ab.txt

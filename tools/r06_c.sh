#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=25 -p no:cacheprovider > gpurun_out/r06_tests_c.log 2>&1
tail -40 gpurun_out/r06_tests_c.log
SECONDS=0
timeout 900 python bench.py > gpurun_out/r06_bench_c.json 2> gpurun_out/r06_bench_c.err
echo "bench rc=$? wall=${SECONDS}s"
python tools/bench_summary.py gpurun_out/r06_bench_c.json || tail -20 gpurun_out/r06_bench_c.err

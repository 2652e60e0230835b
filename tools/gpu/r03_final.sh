#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/r03_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r03_smoke.log
cat gpurun_out/r03_gpu_tests.log gpurun_out/r03_smoke.log

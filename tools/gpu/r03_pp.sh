#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_pp.log; : > $O
timeout 900 python -m pytest tests/test_gpu_postproc.py -x -q 2>&1 | tail -8 >> $O
for a in "32 80 2 8" "32 80 5 40" "32 80 5 40 quant" "64 164 2 8" "64 164 5 40" "64 164 0 0 noise" "32 80 0 0 noise" "2 1000 2 8" "2 1000 5 40"; do
  timeout 200 python tools/pp_bench.py $a 2>&1 | grep separate >> $O
done
echo "-- HVN_WS_BITMAP=0" >> $O
for a in "32 80 5 40" "64 164 0 0 noise"; do
  HVN_WS_BITMAP=0 timeout 200 python tools/pp_bench.py $a 2>&1 | grep separate >> $O
done
cat $O

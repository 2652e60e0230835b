#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_pp3.log; : > $O
for a in "64 164 0 0 noise" "64 164 5 40"; do
  rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p -- python tools/pp_bench.py $a > /tmp/pp.log 2>&1
  grep separate /tmp/pp.log >> $O
  python tools/kernel_stats.py $(find /tmp/prof -name "*_results.db" | head -1) 2>/dev/null | head -14 >> $O
done
cat $O

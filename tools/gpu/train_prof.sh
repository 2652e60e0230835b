#!/bin/bash
# rocprofv3 kernel summaries + executed-FLOP roofline of the training step, per phase.  Outputs: gpurun_out/r03b_train_*.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ph in 0 1; do
  timeout 150 rocprofv3 --kernel-trace --stats -d gpurun_out/r03b_tprof$ph -o t -- python tools/train_bench.py --steps 4 --warmup 2 --phase $ph > gpurun_out/r03b_tb$ph.jsonl 2>gpurun_out/r03b_tprof$ph.err
  db=$(find gpurun_out/r03b_tprof$ph -name "*_results.db" | head -1)
  python tools/kernel_stats.py $db "rocprofv3 --kernel-trace --stats -- python tools/train_bench.py --steps 4 --warmup 2 --phase $ph" > gpurun_out/r03b_train_kernel_stats_phase$ph.csv 2>/dev/null
  tail -1 gpurun_out/r03b_tb$ph.jsonl > gpurun_out/r03b_tb$ph.last
  python tools/train_roofline.py $db gpurun_out/r03b_tb$ph.last 6 > gpurun_out/r03b_train_roofline_phase$ph.json 2>>gpurun_out/r03b_tprof$ph.err
  rm -rf gpurun_out/r03b_tprof$ph
done
cat gpurun_out/r03b_train_roofline_phase0.json gpurun_out/r03b_train_roofline_phase1.json
head -12 gpurun_out/r03b_train_kernel_stats_phase1.csv

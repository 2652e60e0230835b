#!/bin/bash
# One gpurun call: training tests, then the training step with / without the measured conv tiles and the fused BN finalize, then the
# weight-gradient split sweep.  Outputs: gpurun_out/r03_train_tune_*.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -15 > gpurun_out/r03_train_tune_tests.log
{
  echo "# HVN_TILE_SELECT=0 HVN_BN_SPLIT_FINAL=1 (round-2 launch shapes)"
  HVN_TILE_SELECT=0 HVN_BN_SPLIT_FINAL=1 timeout 200 python tools/train_bench.py --steps 8 --warmup 3 2>&1 | tail -2
  echo "# HVN_TILE_SELECT=0 (fused BN finalize only)"
  HVN_TILE_SELECT=0 timeout 200 python tools/train_bench.py --steps 8 --warmup 3 2>&1 | tail -2
  echo "# default (measured conv tiles + fused BN finalize)"
  timeout 200 python tools/train_bench.py --steps 8 --warmup 3 2>&1 | tail -2
} > gpurun_out/r03_train_tune_ab.jsonl
timeout 300 python tools/wgrad_sweep.py --reps 5 > gpurun_out/r03_train_tune_wgrad_sweep.txt 2>&1
cat gpurun_out/r03_train_tune_tests.log
cut -c1-330 gpurun_out/r03_train_tune_ab.jsonl
cat gpurun_out/r03_train_tune_wgrad_sweep.txt

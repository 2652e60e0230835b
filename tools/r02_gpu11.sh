#!/bin/bash
# round-2 GPU call 11: where does the time of short-K conv launches go?  (ablation series per layer shape)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HVN_TILE_SELECT=0
for shape in "32 66 256 1024 1 res" "32 264 64 256 1 res" "32 33 512 2048 1 res" "32 66 1024 256 1 pre" "32 62 1024 256 5" "32 66 256 256 3"; do
  for abl in 0 5 4 3 2 1; do
    HVN_CONV_ABLATE=$abl timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" >> gpurun_out/g11_ablate.log
  done
done
cat gpurun_out/g11_ablate.log

#!/bin/bash
# timing experiment: dense 1x1 conv operands addressed channel-blocked ([C/B][H][W][B], B = 32 or 128) instead of channels-last (results garbage, time only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "0 32" "2 128" "3 128"; do
  set -- $cfg
  HVN_EXP_BLOCKED=$1 HVN_EXP_BLOCK=$2 timeout 120 python tools/layer_ms.py > gpurun_out/g27_layers_blk$1_$2.txt 2>&1
  echo "blocked=$1 block=$2 $(tail -1 gpurun_out/g27_layers_blk$1_$2.txt)"
done
HVN_TILE_SELECT=0 HVN_CONV_ABLATE=6 timeout 100 python tools/conv_bench.py 32 66 256 1024 1 res 2>&1 | grep abl=
HVN_TILE_SELECT=0 HVN_CONV_ABLATE=5 timeout 100 python tools/conv_bench.py 32 66 256 1024 1 res 2>&1 | grep abl=

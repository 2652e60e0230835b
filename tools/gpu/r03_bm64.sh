#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_bm64.log; : > $O
for bm in 128 64 ""; do
  echo "== HVN_CHAIN_BM=$bm" >> $O
  HVN_CHAIN_BM=$bm timeout 200 python tools/layer_ms.py > gpurun_out/r03_layers_bm.txt 2>&1; tail -1 gpurun_out/r03_layers_bm.txt >> $O
  grep -E "conv3\+" gpurun_out/r03_layers_bm.txt >> $O
done
cat $O

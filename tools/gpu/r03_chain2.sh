#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_chain2.log; : > $O
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -15 >> $O
for args in "32 264 64 256 64" "32 264 64 256 64 --x2 64 1" "32 264 64 256 128 --post" "32 132 128 512 128" "32 132 128 512 128 --x2 256 2"; do
  timeout 120 python tools/chain_bench.py $args --unfused 2>&1 | grep -v amdgpu.ids >> $O
  HVN_CHAIN_TRACE=/tmp/ct.bin timeout 120 python tools/chain_bench.py $args 2>&1 | grep -v amdgpu.ids >> $O
  timeout 120 python tools/chain_bench.py $args 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O

#!/bin/bash
# round-2 GPU call 14: per-workgroup timeline of short-K conv launches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for st in 0 1; do
for shape in "32 66 256 1024 1 res" "32 264 64 256 1 res" "32 66 256 256 3"; do
    echo "== stagger=$st $shape" >> gpurun_out/g14_trace.log
    HVN_TILE_SELECT=0 HVN_STAGGER=$st HVN_CONV_TRACE=/tmp/trace.bin timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" >> gpurun_out/g14_trace.log
    python tools/conv_trace.py /tmp/trace.bin >> gpurun_out/g14_trace.log 2>&1
done; done
cat gpurun_out/g14_trace.log

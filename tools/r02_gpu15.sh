#!/bin/bash
# round-2 GPU call 15: epilogue experiments with per-workgroup timelines (5 = baseline, 6 = contiguous tile stores, 7 = all residual loads up front)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for abl in 5 6 7 4; do
for shape in "32 66 256 1024 1 res" "32 264 64 256 1 res" "32 66 1024 256 1 pre" "32 66 256 256 3"; do
    echo "== abl=$abl $shape" >> gpurun_out/g15_trace.log
    HVN_TILE_SELECT=0 HVN_STAGGER=0 HVN_CONV_ABLATE=$abl HVN_CONV_TRACE=/tmp/trace.bin timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" >> gpurun_out/g15_trace.log
    python tools/conv_trace.py /tmp/trace.bin 2>&1 | head -1 >> gpurun_out/g15_trace.log
    HVN_TILE_SELECT=0 HVN_STAGGER=0 HVN_CONV_ABLATE=$abl timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" | sed 's/^/untraced: /' >> gpurun_out/g15_trace.log
done; done
cat gpurun_out/g15_trace.log

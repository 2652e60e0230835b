#!/bin/bash
# round-2 GPU call 18: per-workgroup timelines with the three-phase epilogue (production instantiations) + ablations 4/6 on top of it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for abl in 0 5 6 4; do
for shape in "32 66 256 1024 1 res" "32 264 64 256 1 res" "32 66 1024 256 1 pre"; do
    echo "== abl=$abl $shape" >> gpurun_out/g18_trace.log
    HVN_TILE_SELECT=0 HVN_STAGGER=0 HVN_CONV_ABLATE=$abl HVN_CONV_TRACE=/tmp/trace.bin timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" | sed 's/^/traced:   /' >> gpurun_out/g18_trace.log
    python tools/conv_trace.py /tmp/trace.bin 2>&1 | head -1 >> gpurun_out/g18_trace.log
    HVN_TILE_SELECT=0 HVN_STAGGER=0 HVN_CONV_ABLATE=$abl timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" | sed 's/^/untraced: /' >> gpurun_out/g18_trace.log
done; done
cat gpurun_out/g18_trace.log

#!/bin/bash
# round-2 GPU call 13: stagger variants (LDS-base priority, slot priority, delayed start)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for st in 0 1 2 3; do
for shape in "32 66 256 1024 1 res" "32 264 64 256 1 res" "32 66 1024 256 1 pre" "32 66 256 256 3"; do
    HVN_TILE_SELECT=0 HVN_STAGGER=$st timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" | sed "s/^/stagger=$st /" >> gpurun_out/g13_stagger.log
done; done
timeout 600 python -m pytest "tests/test_gpu_train.py::test_train_mode_forward_is_a_torch_autograd_node" -q -m gpu -x -s 2>&1 | grep -E "autograd path|assert|Error|passed|failed" | head -12 >> gpurun_out/g13_stagger.log
cat gpurun_out/g13_stagger.log

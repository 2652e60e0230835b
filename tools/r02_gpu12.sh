#!/bin/bash
# round-2 GPU call 12: staggering the two co-resident workgroups by issue priority
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for st in 1 0; do
for shape in "32 66 256 1024 1 res" "32 264 64 256 1 res" "32 66 1024 256 1 pre" "32 66 256 256 3" "32 62 1024 256 5"; do
    HVN_TILE_SELECT=0 HVN_STAGGER=$st timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" | sed "s/^/stagger=$st /" >> gpurun_out/g12_stagger.log
done; done
for st in 1 0; do
HVN_STAGGER=$st timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('stagger=$st', 'tiles/s %.1f' % r['value'], 'network_ms %.2f' % r['config']['stage_ms']['network'], 'conv_ms %.2f' % r['roofline']['conv_ms_per_step'], 'frac %.4f' % r['roofline']['frac'])
" >> gpurun_out/g12_stagger.log 2>&1
done
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_net.py::test_forward_matches_reference_golden "tests/test_gpu_train.py::test_train_mode_forward_is_a_torch_autograd_node" -q -m gpu -x 2>&1 | tail -5 >> gpurun_out/g12_stagger.log
cat gpurun_out/g12_stagger.log

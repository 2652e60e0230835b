#!/bin/bash
# round-2 GPU call 20: raised issue priority for the epilogue (HVN_STAGGER=9 switches it off)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for st in 0 9; do
for shape in "32 66 256 1024 1 res" "32 264 64 256 1 res" "32 66 1024 256 1 pre" "32 66 256 256 3"; do
    HVN_TILE_SELECT=0 HVN_STAGGER=$st timeout 120 python tools/conv_bench.py $shape 2>&1 | grep "abl=" | sed "s/^/epilogue_prio=$([ $st = 0 ] && echo on || echo off) /" >> gpurun_out/g20.log
done
HVN_STAGGER=$st timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('HVN_STAGGER=$st', 'tiles/s %.1f' % r['value'], 'network_ms %.2f' % r['config']['stage_ms']['network'], 'conv_ms %.2f' % r['roofline']['conv_ms_per_step'], 'frac %.4f' % r['roofline']['frac'])
" >> gpurun_out/g20.log 2>&1
done
cat gpurun_out/g20.log

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_bf16_chain_bench.log; : > $O
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bench_shapes.py -q -x -p no:cacheprovider 2>&1 | tail -4 >> $O
A="--dtype bf16 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-traffic"
for c in 0 d0d1 0 d0d1; do
  HVN_BF16_CHAIN=$c timeout 400 python bench.py $A 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d.get('roofline', {})
print('HVN_BF16_CHAIN=$c value %.1f tiles/s ms_per_step %.2f | conv_ms %.2f frac %.4f launches %s' % (d['value'], d['ms_per_step'], r.get('conv_ms_per_step', 0), r.get('frac', 0), r.get('timed_launches_per_step')))" >> $O 2>&1
done
cat $O

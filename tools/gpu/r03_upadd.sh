#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_upadd.log; : > $O
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_conv.py tests/test_gpu_net.py tests/test_gpu_bench_shapes.py -x -q 2>&1 | tail -4 >> $O
for f in 1 0; do
  HVN_FUSE_UPADD=$f timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('HVN_FUSE_UPADD=$f tiles/s %.1f ms/step %.2f conv_ms %.2f frac %.4f timed launches %d' % (d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], r['timed_launches_per_step']))" >> $O
done
cat $O

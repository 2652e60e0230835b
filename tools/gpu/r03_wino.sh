#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_wino.log; : > $O
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_chain.py -x -q 2>&1 | tail -4 >> $O
for cfg in "4 4" "6 4" "4 6" "6 6"; do
  set -- $cfg
  echo "== HVN_WINOGRAD3_M=$1 HVN_WINOGRAD=$2" >> $O
  HVN_WINOGRAD3_M=$1 HVN_WINOGRAD=$2 timeout 300 python -m pytest tests/test_gpu_net.py -x -q -k "golden" -s 2>&1 | grep -E "passed|failed|err|Error" | tail -4 >> $O
  HVN_WINOGRAD3_M=$1 HVN_WINOGRAD=$2 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tiles/s %.1f conv_ms %.2f frac %.4f exec_gflop %.0f' % (d['value'], r['conv_ms_per_step'], r['frac'], r['executed_gflop_per_step']))" >> $O
done
cat $O

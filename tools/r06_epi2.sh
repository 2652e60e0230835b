#!/bin/bash
# round 6: conv epilogue addressing by stepped 32-bit offsets (this build) vs HEAD~'s library (variant "prev": 64-bit address products), one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_epilogue_addr_ab.log; : > $O
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_x3.py tests/test_gpu_chain.py tests/test_gpu_bf16.py tests/test_gpu_net.py tests/test_gpu_bench_shapes.py -q -x -p no:cacheprovider 2>&1 | tail -3 >> $O
for v in prev ""; do
  f=gpurun_out/r06_layers_addr_${v:-new}.txt
  HVN_LIB_VARIANT=$v timeout 300 python tools/layer_ms.py 2>/dev/null > $f; echo "== fp32 cfg 2, HVN_LIB_VARIANT=$v: $(tail -1 $f)" >> $O
  f=gpurun_out/r06_layers_cfg3_addr_${v:-new}.txt
  HVN_LIB_VARIANT=$v timeout 300 python tools/layer_ms.py --dtype bf16 --mode fast --nr-types 6 --batch 64 2>/dev/null > $f; echo "== bf16 cfg 3, HVN_LIB_VARIANT=$v: $(tail -1 $f)" >> $O
done
Q="--steps 20 --no-cpu-baseline --no-variants --no-traffic --checkpoint random --no-roofline"
for v in prev "" prev ""; do
  HVN_LIB_VARIANT=$v timeout 300 python bench.py $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg 2 HVN_LIB_VARIANT=$v value %.1f ms_per_step %.2f' % (d['value'], d['ms_per_step']))" >> $O 2>&1
done
A="--dtype bf16 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-traffic --no-roofline"
for v in prev "" prev ""; do
  HVN_LIB_VARIANT=$v timeout 400 python bench.py $A 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg 3 HVN_LIB_VARIANT=$v value %.1f ms_per_step %.2f' % (d['value'], d['ms_per_step']))" >> $O 2>&1
done
cat $O

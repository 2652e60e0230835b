#!/bin/bash
# round 6: csrc/hvn_conv_chain_bf16.hip -- bit-equality tests (short timeout), per-launch tables and cfg-3 bench lines for HVN_BF16_CHAIN = 0 | d0 | d0d1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_bf16_chain.log; : > $O
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -x -p no:cacheprovider -k "chained_seams" --tb=short 2>&1 | tail -15 >> $O
A="--dtype bf16 --mode fast --nr-types 6 --batch 64"
for c in 0 d0 d0d1; do
  f=gpurun_out/r06_layers_cfg3_chain_$c.txt
  HVN_BF16_CHAIN=$c timeout 300 python tools/layer_ms.py $A 2>/dev/null | grep -v amdgpu.ids > $f; echo "== HVN_BF16_CHAIN=$c: $(tail -1 $f)" >> $O
  grep -E "^d0|^d1" $f | head -26 >> $O
done
Q="$A --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-traffic --checkpoint random --no-roofline"
for c in 0 d0 d0d1 0 d0d1; do
  HVN_BF16_CHAIN=$c timeout 300 python bench.py $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HVN_BF16_CHAIN=$c value %.1f ms_per_step %.2f' % (d['value'], d['ms_per_step']))" >> $O 2>&1
done
cat $O

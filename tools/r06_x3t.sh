#!/bin/bash
# round 6: hvn_conv_igemm_x3t (128 x 128, ring of two 16-deep half stages, 40 KB LDS, <= 168 VGPRs: THREE workgroups per CU) offered to the
# engine's timing pass (default) vs not offered (HVN_X3T=0), one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_x3t_ab.log; : > $O
timeout 900 python -m pytest tests/test_gpu_x3.py -q -x -p no:cacheprovider -k "lds_dma" 2>&1 | tail -3 >> $O
for v in "HVN_X3T=0" "HVN_X3G_FORCE=1664" "HVN_X3T=1"; do
  f=gpurun_out/r06_layers_x3t_${v//=/_}.txt
  env $v timeout 300 python tools/layer_ms.py 2>/dev/null > $f; echo "== $v: $(tail -1 $f)" >> $O
done
Q="--steps 20 --no-cpu-baseline --no-variants --no-traffic --checkpoint random --no-roofline"
for v in 0 1 0 1; do
  HVN_X3T=$v timeout 300 python bench.py $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg 2 HVN_X3T=$v value %.1f ms_per_step %.2f' % (d['value'], d['ms_per_step']))" >> $O 2>&1
done
cat $O

#!/bin/bash
# round 6: hvn_conv_igemm_x3g wave tile 32 x 128 (A split once per wave; default) vs round 5's 64 x 64 (library variant "wn1" = the 32 x 128 tile), one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_x3g_wave_tile_ab.log; : > $O
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_net.py -q -x -p no:cacheprovider 2>&1 | tail -3 >> $O
for v in "" wn1; do
  f=gpurun_out/r06_layers_wavetile_${v:-32x128}.txt
  HVN_LIB_VARIANT=$v timeout 300 python tools/layer_ms.py 2>/dev/null > $f; echo "== HVN_LIB_VARIANT=$v: $(tail -1 $f)" >> $O
done
Q="--steps 20 --no-cpu-baseline --no-variants --no-traffic --checkpoint random --no-roofline"
for v in "" wn1 "" wn1; do
  HVN_LIB_VARIANT=$v timeout 300 python bench.py $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg 2 HVN_LIB_VARIANT=$v value %.1f ms_per_step %.2f' % (d['value'], d['ms_per_step']))" >> $O 2>&1
done
cat $O

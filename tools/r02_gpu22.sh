#!/bin/bash
# A/B of the swizzled-LDS build (libhvn_hip_swz.so: 3 workgroups of 128x64 tiles per CU) against the default build, one box.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/g22
( timeout 300 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
  HVN_LIB_VARIANT=swz timeout 300 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
  HVN_LIB_VARIANT=swz HVN_FORCE_TILE_N=64 timeout 300 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
  HVN_LIB_VARIANT=swz HVN_NO_DENSE_KERNEL=1 timeout 300 python -m pytest tests/test_gpu_conv.py -q -x -k dense 2>&1 | tail -2
  HVN_LIB_VARIANT=swz timeout 400 python -m pytest tests/test_gpu_net.py -q -x -k "golden or batch_matches" 2>&1 | tail -2 ) > ${O}_tests.log 2>&1
for v in base swz; do for f in 128 64; do
  V=""; [ $v = swz ] && V=swz
  HVN_LIB_VARIANT=$V HVN_FORCE_TILE_N=$f timeout 200 python tools/layer_ms.py > ${O}_layers_${v}_${f}.txt 2>&1
done; done
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants"
run() { tag=$1; shift; env "$@" timeout 200 $B 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$tag', 'tiles/s %.1f conv_ms %.2f frac %.4f network_ms %.2f' % (d['value'], r['conv_ms_per_step'], r['frac'], d['config'].get('network_ms', -1)))" >> ${O}_bench.log 2>&1; }
run base_default HVN_X=0
run swz_slots512 HVN_LIB_VARIANT=swz
run swz_768_c054 HVN_LIB_VARIANT=swz HVN_WG_SLOTS_64=768 HVN_NARROW_COST=0.54
run swz_768_c045 HVN_LIB_VARIANT=swz HVN_WG_SLOTS_64=768 HVN_NARROW_COST=0.45
run swz_768_c038 HVN_LIB_VARIANT=swz HVN_WG_SLOTS_64=768 HVN_NARROW_COST=0.38
cat ${O}_tests.log ${O}_bench.log; for f in ${O}_layers_*.txt; do tail -1 $f; done

#!/bin/bash
# First GPU call of the next session: A/B of the two epilogue variants prepared at the end of round 2 (hvn_conv.hip HVN_EPI_LINEAR / HVN_NT).
# Before calling:  python -c "from hover_net_amd import lib; [lib.build_variant(v) for v in ('lin', 'nt', 'lin_nt', 'trace')]"   (the .so files travel with the snapshot)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in lin nt lin_nt; do
  [ -f hover_net_amd/libhvn_hip_$v.so ] || continue
  echo "== $v" >> gpurun_out/r03_first.log
  HVN_LIB_VARIANT=$v timeout 120 python -m pytest tests/test_gpu_conv.py tests/test_gpu_net.py -q -x -k "conv or golden" 2>&1 | tail -2 >> gpurun_out/r03_first.log
  HVN_LIB_VARIANT=$v timeout 60 python tools/layer_ms.py > gpurun_out/r03_layers_$v.txt 2>&1; tail -1 gpurun_out/r03_layers_$v.txt >> gpurun_out/r03_first.log
done
timeout 60 python tools/layer_ms.py > gpurun_out/r03_layers_default.txt 2>&1; tail -1 gpurun_out/r03_layers_default.txt >> gpurun_out/r03_first.log
# where a short-K tile's epilogue spends its time, on the PRODUCTION instantiations (build variant `trace`)
if [ -f hover_net_amd/libhvn_hip_trace.so ]; then
  for shape in "32 66 256 1024 1 res" "32 264 64 256 1 res" "32 33 512 2048 1 res" "32 66 1024 256 1 pre"; do
    HVN_LIB_VARIANT=trace HVN_TILE_SELECT=0 HVN_CONV_TRACE=/tmp/trace.bin timeout 100 python tools/conv_bench.py $shape 2>&1 | grep "abl=" >> gpurun_out/r03_first.log
    python tools/conv_trace.py /tmp/trace.bin --fine >> gpurun_out/r03_first.log 2>&1
  done
fi
cat gpurun_out/r03_first.log

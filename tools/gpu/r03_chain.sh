#!/bin/bash
# OP_CHAIN bring-up: parity tests, per-launch table and bench line with the chains on / off, on one box.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_chain.log; : > $O
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -15 >> $O
for ch in 1 0; do
  HVN_CHAIN=$ch timeout 200 python tools/layer_ms.py > gpurun_out/r03_layers_chain$ch.txt 2>&1; tail -1 gpurun_out/r03_layers_chain$ch.txt >> $O
  HVN_CHAIN=$ch timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/r03_bench_chain$ch.json 2>> $O
  python -c "import json;d=json.load(open('gpurun_out/r03_bench_chain$ch.json'));print('chain=$ch', d['value'], d['roofline']['conv_ms_per_step'], d['roofline']['frac'])" >> $O 2>&1
done
for v in lin lin_nt; do
  [ -f hover_net_amd/libhvn_hip_$v.so ] || continue
  HVN_LIB_VARIANT=$v timeout 200 python tools/layer_ms.py > gpurun_out/r03_layers_$v.txt 2>&1; tail -1 gpurun_out/r03_layers_$v.txt >> $O
done
cat $O

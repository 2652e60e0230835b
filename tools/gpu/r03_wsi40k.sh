#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wsi_merge.py tests/test_gpu_net.py -x -q -k "merge or wsi" 2>&1 | tail -3
for lanes in 1 2 3; do
HVN_WSI_LANES=$lanes timeout 900 python tools/wsi_bench.py --size 16384 --skip-stage1 2>&1 | tail -1 | cut -c1-600
done
timeout 900 python tools/wsi_bench.py --size 40000 --skip-stage1 2>&1 | tail -1 > gpurun_out/r03_wsi_40k_stage2.json
cut -c1-900 gpurun_out/r03_wsi_40k_stage2.json

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r03_wsi.log; : > $O
timeout 900 python -m pytest tests/test_gpu_wsi_merge.py tests/test_gpu_net.py -x -q -k "merge or wsi" 2>&1 | tail -12 >> $O
HVN_WSI_HOST_MERGE=1 timeout 600 python tools/wsi_bench.py --size 8192 --skip-stage1 2>&1 | tail -1 | cut -c1-700 >> $O
timeout 600 python tools/wsi_bench.py --size 8192 --skip-stage1 2>&1 | tail -1 | cut -c1-700 >> $O
cat $O

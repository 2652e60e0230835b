#!/bin/bash
# round-2 GPU call 5: BASELINE cfg 4 at its stated size on ONE GPU (40 000^2 synthetic slide: 248 004 patches, 400+760+361 tiles)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/g5_prof -o r -- python tools/wsi_bench.py --size 4096 --skip-stage1 > gpurun_out/g5_prof.log 2>&1
python tools/kernel_stats.py gpurun_out/g5_prof/r_results.db "wsi_bench.py --size 4096 --skip-stage1 (stage 2 only)" > gpurun_out/g5_stage2_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/g5_prof
timeout 1100 python tools/wsi_bench.py --size 40000 > gpurun_out/g5_wsi40k.log 2>&1
tail -1 gpurun_out/g5_wsi40k.log; head -16 gpurun_out/g5_stage2_kernel_stats.csv

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; R=r03
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-traffic > gpurun_out/${R}_bench_profiled_run.json 2>gpurun_out/${R}_prof.err
db=$(find gpurun_out/${R}_prof -name "*_results.db" | head -1)
python tools/kernel_stats.py $db "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-traffic" > gpurun_out/${R}_kernel_stats_bench_b32.csv 2>/dev/null
python tools/layer_table.py $db 32 > gpurun_out/${R}_conv_layer_table.txt 2>/dev/null
rm -rf gpurun_out/${R}_prof
head -4 gpurun_out/${R}_kernel_stats_bench_b32.csv | cut -c1-160

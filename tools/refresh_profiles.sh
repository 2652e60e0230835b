#!/bin/bash
# Round-end refresh on the GPU box (one gpurun call):  gpurun -- 'bash tools/refresh_profiles.sh r04'
# full -m gpu suite (TESTS=0 skips it), smoke(), the default bench line (fitted checkpoint; measures roofline.traffic itself through two
# rocprofv3 --pmc child passes and carries the cfg-3 leg), rocprofv3 kernel stats + per-layer table of a random-checkpoint run (a fit
# under the tracer would add 300 k training launches), the SQ PMC pass, per-launch tables (fp32 pipe | bf16x3), cfg 3 as its own line +
# table, optionally the 40 000^2 whole-slide run (WSI40K=1, ~8 min).  Outputs land in gpurun_out/; copy what is to be judged to profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=${1:-r}
if [ "${TESTS:-1}" != "0" ]; then
  timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/${R}_gpu_tests.log
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${R}_smoke.log
timeout 900 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
PCMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_prof -o r -- $PCMD > gpurun_out/${R}_bench_profiled_run.json 2>gpurun_out/${R}_prof.err
db=$(find gpurun_out/${R}_prof -name "*_results.db" | head -1)
python tools/kernel_stats.py $db "rocprofv3 --kernel-trace --stats -- $PCMD" > gpurun_out/${R}_kernel_stats_bench_b32.csv 2>/dev/null
python tools/layer_table.py $db 32 > gpurun_out/${R}_conv_layer_table.txt 2>/dev/null
rm -rf gpurun_out/${R}_prof
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d gpurun_out/${R}_pmcS -o p -- python bench.py --pmc-child > gpurun_out/${R}_pmcS.log 2>&1
python tools/pmc_sq.py $(find gpurun_out/${R}_pmcS -name "*_results.db" | head -1) gpurun_out/${R}_pmc_sq_conv.json > /dev/null 2>gpurun_out/${R}_pmcS.err
rm -rf gpurun_out/${R}_pmcS
for x in 0 6; do HVN_X3=$x timeout 300 python tools/layer_ms.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_layers_x3_$x.txt; done
timeout 400 python bench.py --dtype bf16 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-traffic > gpurun_out/${R}_bench_cfg3_fast_b64_bf16.json 2>/dev/null
timeout 200 python tools/layer_ms.py --dtype bf16 --mode fast --nr-types 6 --batch 64 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_layers_cfg3_bf16.txt
if [ -n "$WSI40K" ]; then
  timeout 1200 python tools/wsi_bench.py --size 40000 2>&1 | tail -1 > gpurun_out/${R}_wsi_40k.json
fi
cat gpurun_out/${R}_gpu_tests.log gpurun_out/${R}_smoke.log 2>/dev/null | tail -8
python tools/bench_summary.py gpurun_out/${R}_bench.json
tail -1 gpurun_out/${R}_layers_x3_0.txt gpurun_out/${R}_layers_x3_6.txt gpurun_out/${R}_layers_cfg3_bf16.txt
cut -c1-300 gpurun_out/${R}_wsi_40k.json 2>/dev/null
cat gpurun_out/${R}_pmc_sq_conv.json 2>/dev/null | head -12

#!/bin/bash
# Round-end refresh on the GPU box (one gpurun call): full -m gpu suite, smoke(), the default bench line, rocprofv3 kernel stats +
# per-layer table, the three PMC passes (separate runs: FETCH_SIZE, WRITE_SIZE, SQ), the cfg-3 lines and the training bench.
# Outputs land in gpurun_out/; tools/{kernel_stats,pmc_traffic,pmc_sq,layer_table}.py reduce them; copy the results to profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=${1:-r}
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/${R}_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${R}_smoke.log
timeout 600 python bench.py > gpurun_out/${R}_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/${R}_prof.log 2>&1
python tools/kernel_stats.py gpurun_out/${R}_prof/r_results.db "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants" > gpurun_out/${R}_kernel_stats.csv 2>/dev/null
python tools/layer_table.py gpurun_out/${R}_prof/r_results.db 32 > gpurun_out/${R}_layer_table.txt 2>/dev/null
rm -rf gpurun_out/${R}_prof
timeout 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${R}_pmcF -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/${R}_pmcF.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${R}_pmcW -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/${R}_pmcW.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d gpurun_out/${R}_pmcS -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/${R}_pmcS.log 2>&1
python tools/pmc_traffic.py gpurun_out/${R}_pmcF/p_results.db gpurun_out/${R}_pmcW/p_results.db gpurun_out/${R}_pmc_traffic.json > /dev/null 2>gpurun_out/${R}_pmcT.err
python tools/pmc_sq.py gpurun_out/${R}_pmcS/p_results.db gpurun_out/${R}_pmc_sq_conv.json > /dev/null 2>gpurun_out/${R}_pmcS.err
rm -rf gpurun_out/${R}_pmcF gpurun_out/${R}_pmcW gpurun_out/${R}_pmcS
if [ -z "$SKIP_CFG3" ]; then
timeout 300 python bench.py --dtype bf16 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${R}_bench_cfg3_bf16.log 2>&1
timeout 300 python bench.py --dtype fp32 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${R}_bench_cfg3_fp32.log 2>&1
fi
if [ -z "$SKIP_TRAIN" ]; then
timeout 300 python tools/train_bench.py --steps 8 --warmup 3 > gpurun_out/${R}_train_bench.jsonl 2> gpurun_out/${R}_train_bench.err
fi
# A/B of the LDS layout on the same box: the padded-row build (two 128x64 workgroups per CU) with the rounds model, and per-layer times of both
if [ -f hover_net_amd/libhvn_hip_pad.so ]; then
  HVN_LIB_VARIANT=pad HVN_TILE_SELECT=model HVN_WG_SLOTS_64=512 HVN_NARROW_COST=0.54 timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants 2>&1 | tail -1 > gpurun_out/${R}_bench_padded_lds.log
  timeout 200 python tools/layer_ms.py > gpurun_out/${R}_layers_swizzled_autotuned.txt 2>&1
  HVN_LIB_VARIANT=pad HVN_TILE_SELECT=model HVN_WG_SLOTS_64=512 HVN_NARROW_COST=0.54 timeout 200 python tools/layer_ms.py > gpurun_out/${R}_layers_padded_model.txt 2>&1
fi
cat gpurun_out/${R}_gpu_tests.log gpurun_out/${R}_smoke.log
for f in ${R}_bench ${R}_bench_cfg3_bf16 ${R}_bench_cfg3_fp32 ${R}_bench_padded_lds; do [ -f gpurun_out/$f.log ] && tail -1 gpurun_out/$f.log | cut -c1-200; done
tail -1 gpurun_out/${R}_layers_swizzled_autotuned.txt gpurun_out/${R}_layers_padded_model.txt 2>/dev/null
cat gpurun_out/${R}_pmc_traffic.json gpurun_out/${R}_pmc_sq_conv.json 2>/dev/null | head -30; cat gpurun_out/${R}_train_bench.jsonl | cut -c1-200

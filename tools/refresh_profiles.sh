#!/bin/bash
# Round-end refresh on the GPU box (one gpurun call):  gpurun -- 'bash tools/refresh_profiles.sh r06'
# full -m gpu suite (TESTS=0 skips it), run-to-run / form-to-form bit equality of the whole network under load, smoke(), the default bench line (fitted checkpoint; measures roofline.traffic itself through two
# rocprofv3 --pmc child passes, writes the per-launch-class traffic table of those passes, carries the cfg-3 / train_step / wsi_8k legs),
# rocprofv3 kernel stats + per-layer table of a random-checkpoint run on ONE launch stream (every kernel's duration is its own: under the
# two-stream schedule overlapped kernels' durations inflate), the SQ PMC pass, per-launch tables (fp32 pipe | default), cfg 3 as its own
# line + table, the training step's kernel summaries + roofline per phase, optionally the 40 000^2 whole-slide run (WSI40K=1, ~7 min).
# Outputs land in gpurun_out/; copy what is to be judged to profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=${1:-r}
if [ "${TESTS:-1}" != "0" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -30 > gpurun_out/${R}_gpu_tests.log
fi
timeout 400 python tools/determinism_check.py 16 3 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_determinism.txt      # every kernel form, whole network, batch 16: same bits
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${R}_smoke.log
HVN_KEEP_PMC_TABLE=gpurun_out/${R}_traffic_by_kernel.txt timeout 900 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
PCMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-traffic --checkpoint random"
HVN_SPLIT=1 HVN_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_prof -o r -- $PCMD > gpurun_out/${R}_bench_profiled_run.json 2>gpurun_out/${R}_prof.err
db=$(find gpurun_out/${R}_prof -name "*_results.db" | head -1)
python tools/kernel_stats.py $db "HVN_SPLIT=1 HVN_LANES=0 rocprofv3 --kernel-trace --stats -- $PCMD   (ONE launch stream: every plan execution of this process is single-stream)" > gpurun_out/${R}_kernel_stats_single_stream.csv 2>/dev/null
python tools/layer_table.py $db 32 > gpurun_out/${R}_conv_layer_table.txt 2>/dev/null
rm -rf gpurun_out/${R}_prof
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d gpurun_out/${R}_pmcS -o p -- python bench.py --pmc-child > gpurun_out/${R}_pmcS.log 2>&1
python tools/pmc_sq.py $(find gpurun_out/${R}_pmcS -name "*_results.db" | head -1) gpurun_out/${R}_pmc_sq_conv.json > /dev/null 2>gpurun_out/${R}_pmcS.err
rm -rf gpurun_out/${R}_pmcS
HVN_X3=0 timeout 300 python tools/layer_ms.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_layers_fp32_pipe.txt
timeout 300 python tools/layer_ms.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_layers_default.txt
timeout 400 python bench.py --dtype bf16 --mode fast --nr-types 6 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-traffic > gpurun_out/${R}_bench_cfg3_fast_b64_bf16.json 2>/dev/null
timeout 200 python tools/layer_ms.py --dtype bf16 --mode fast --nr-types 6 --batch 64 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_layers_cfg3_bf16.txt
for ph in 0 1; do
  # one stream (round 6's branch / weight-gradient streams off): under concurrency overlapped kernels' durations inflate; the default step is timed right after
  CMD="env HVN_TRAIN_BRANCH_STREAMS=0 HVN_TRAIN_WGRAD_STREAM=0 python tools/train_bench.py --steps 4 --warmup 2 --phase $ph"
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_tprof$ph -o r -- $CMD 2>/dev/null | grep "^{" > gpurun_out/${R}_train_profiled_phase$ph.json
  timeout 300 python tools/train_bench.py --steps 8 --warmup 3 --phase $ph 2>/dev/null | grep "^{" > gpurun_out/${R}_train_default_phase$ph.json
  tdb=$(find gpurun_out/${R}_tprof$ph -name "*_results.db" | head -1)
  python tools/kernel_stats.py $tdb "rocprofv3 --kernel-trace --stats -- $CMD" > gpurun_out/${R}_train_kernel_stats_phase$ph.csv 2>/dev/null
  python tools/train_roofline.py $tdb gpurun_out/${R}_train_profiled_phase$ph.json 6 > gpurun_out/${R}_train_roofline_phase$ph.json 2>/dev/null
  rm -rf gpurun_out/${R}_tprof$ph
done
# cfg 3's declared tolerance as a distribution: 8 deterministic fits x 48 tiles, bf16 vs fp32 segmentation (weights hash per fit: equal on every box)
timeout 900 python tools/bf16_pq_table.py --out gpurun_out/${R}_bf16_pq_table.json 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_bf16_pq_table.txt
if [ -n "$WSI40K" ]; then
  timeout 1200 python tools/wsi_bench.py --size 40000 2>&1 | tail -1 > gpurun_out/${R}_wsi_40k.json
fi
cat gpurun_out/${R}_gpu_tests.log gpurun_out/${R}_smoke.log 2>/dev/null | tail -8
python tools/bench_summary.py gpurun_out/${R}_bench.json
for f in gpurun_out/${R}_layers_fp32_pipe.txt gpurun_out/${R}_layers_default.txt gpurun_out/${R}_layers_cfg3_bf16.txt; do tail -n 1 $f; done
cat gpurun_out/${R}_traffic_by_kernel.txt 2>/dev/null
cat gpurun_out/${R}_train_roofline_phase0.json gpurun_out/${R}_train_roofline_phase1.json 2>/dev/null | cut -c1-400
cut -c1-300 gpurun_out/${R}_wsi_40k.json 2>/dev/null
cat gpurun_out/${R}_pmc_sq_conv.json 2>/dev/null | head -12

#!/bin/bash
# One gpurun call: training tests + smoke on the final state, the step with / without the measured launch shapes, rocprofv3 kernel
# summaries + executed-FLOP roofline per phase.  Outputs: gpurun_out/r03b_*.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -6 > gpurun_out/r03b_train_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03b_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r03b_smoke.log
{
  echo "# HVN_TILE_SELECT=0 (static conv tiles and weight-gradient split)"
  HVN_TILE_SELECT=0 timeout 200 python tools/train_bench.py --steps 8 --warmup 3 2>&1 | tail -2
  echo "# default (conv tiles and weight-gradient split measured per launch shape)"
  timeout 200 python tools/train_bench.py --steps 8 --warmup 3 2>&1 | tail -2
} > gpurun_out/r03b_train_bench.jsonl
for ph in 0 1; do
  timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r03b_tprof$ph -o t -- python tools/train_bench.py --steps 4 --warmup 2 --phase $ph > gpurun_out/r03b_tb$ph.jsonl 2>gpurun_out/r03b_tprof$ph.err
  db=$(find gpurun_out/r03b_tprof$ph -name "*_results.db" | head -1)
  python tools/kernel_stats.py $db "rocprofv3 --kernel-trace --stats -- python tools/train_bench.py --steps 4 --warmup 2 --phase $ph" > gpurun_out/r03b_train_kernel_stats_phase$ph.csv 2>/dev/null
  tail -1 gpurun_out/r03b_tb$ph.jsonl > gpurun_out/r03b_tb$ph.last
  python tools/train_roofline.py $db gpurun_out/r03b_tb$ph.last 6 > gpurun_out/r03b_train_roofline_phase$ph.json 2>>gpurun_out/r03b_tprof$ph.err
  rm -rf gpurun_out/r03b_tprof$ph
done
cat gpurun_out/r03b_train_tests.log; tail -3 gpurun_out/r03b_smoke.log
cut -c1-260 gpurun_out/r03b_train_bench.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r03b_train_bench.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print(d["phase"], round(d["ms_per_step"], 2), d.get("conv_tiles"))
PY
cat gpurun_out/r03b_train_roofline_phase0.json gpurun_out/r03b_train_roofline_phase1.json
head -8 gpurun_out/r03b_train_kernel_stats_phase1.csv

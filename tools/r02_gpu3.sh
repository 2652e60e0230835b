#!/bin/bash
# round-2 GPU call 3: tile-select / stream-mode A/B, drop-in + loss-weight tests, WSI stage-2 breakdown
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_bench_shapes.py tests/test_gpu_conv.py "tests/test_gpu_train.py::test_losses_and_logit_gradients" tests/test_gpu_train.py::test_training_step_matches_oracle tests/test_gpu_train.py::test_two_phase_schedule_runs_and_learns -q -m gpu -x 2>&1 | tail -15 > gpurun_out/g3_tests.log
ab() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$label', 'tiles/s %.1f' % r['value'], 'network_ms %.2f' % r['config']['stage_ms']['network'])
" >> gpurun_out/g3_ab.log 2>&1
}
ab "tilesel=1 split=1 lanes=0" HVN_TILE_SELECT=1
ab "tilesel=0 split=1 lanes=0" HVN_TILE_SELECT=0
ab "tilesel=1 split=1 lanes=2" HVN_LANES=2
ab "tilesel=1 split=2 lanes=0" HVN_SPLIT=2
ab "tilesel=1 split=2 lanes=2" HVN_SPLIT=2 HVN_LANES=2
timeout 600 python tools/wsi_bench.py --size 8192 --skip-stage1 > gpurun_out/g3_wsi.log 2>&1
cat gpurun_out/g3_tests.log gpurun_out/g3_ab.log; tail -2 gpurun_out/g3_wsi.log

#!/bin/bash
# round-2 GPU call 4: depth-first head A/B, pipelined WSI stage 2, changed-path tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ab() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$label', 'tiles/s %.1f' % r['value'], 'network_ms %.2f' % r['config']['stage_ms']['network'], 'conv_ms %.2f' % r['roofline']['conv_ms_per_step'], 'frac %.4f' % r['roofline']['frac'])
" >> gpurun_out/g4_ab.log 2>&1
}
ab "baseline" HVN_DF_UPTO=
ab "df d1 b1" HVN_DF_UPTO=d1. HVN_DF_BATCH=1
ab "df d1 b2" HVN_DF_UPTO=d1. HVN_DF_BATCH=2
ab "df d2 b1" HVN_DF_UPTO=d2. HVN_DF_BATCH=1
ab "df d2 b2" HVN_DF_UPTO=d2. HVN_DF_BATCH=2
ab "df d2 b4" HVN_DF_UPTO=d2. HVN_DF_BATCH=4
ab "df d3 b4" HVN_DF_UPTO=d3. HVN_DF_BATCH=4
timeout 900 python -m pytest tests/test_gpu_net.py::test_wsi_pipeline_on_synthetic_slide tests/test_gpu_net.py::test_process_images_tile_pipeline tests/test_gpu_train.py::test_two_phase_schedule_runs_and_learns tests/test_gpu_postproc.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/g4_tests.log
timeout 600 python tools/wsi_bench.py --size 8192 > gpurun_out/g4_wsi.log 2>&1
cat gpurun_out/g4_ab.log gpurun_out/g4_tests.log; tail -1 gpurun_out/g4_wsi.log

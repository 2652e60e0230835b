#!/bin/bash
# round-2 GPU call 2: persistent conv kernel -- parity tests, A/B timing, per-layer table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_net.py tests/test_gpu_bench_shapes.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/g2_tests.log
for cfg in "2 1" "0 1" "2 0" "1 1" "3 1"; do
  set -- $cfg
  HVN_PERSIST=$1 HVN_RES_PREFETCH=$2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('persist=$1 respf=$2', 'tiles/s %.1f' % r['value'], 'conv_ms %.2f' % r['roofline']['conv_ms_per_step'], 'frac %.4f' % r['roofline']['frac'], 'network_ms %.2f' % r['config']['stage_ms']['network'])
" >> gpurun_out/g2_ab.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/g2_prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/g2_prof.log 2>&1
python tools/kernel_stats.py gpurun_out/g2_prof/r_results.db "bench.py --steps 3 --warmup 1 (persistent conv kernel)" > gpurun_out/g2_kernel_stats.csv 2>gpurun_out/g2_ks.err
python tools/layer_table.py gpurun_out/g2_prof/r_results.db 32 > gpurun_out/g2_layer_table.txt 2>gpurun_out/g2_lt.err
rm -rf gpurun_out/g2_prof
cat gpurun_out/g2_tests.log gpurun_out/g2_ab.log; grep -E "^(d0|d1|d2|d3|conv_bot|decoder.tp.u3.(conva|dense)) " gpurun_out/g2_layer_table.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import os; print('cpus', os.cpu_count(), len(os.sched_getaffinity(0)))"
timeout 1500 python -m pytest tests -q -m gpu -x --durations=30 -p no:cacheprovider > gpurun_out/r06_tests_e.log 2>&1
tail -45 gpurun_out/r06_tests_e.log
for d in 0 1; do
  HVN_TRAIN_ZERO_ASYNC=$d timeout 400 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('HVN_TRAIN_ZERO_ASYNC=$d phase', d['phase'], 'ms/step %.2f' % d['ms_per_step'], 'fwd %.2f bwd %.2f' % (d['forward_ms'], d['loss_backward_ms']))"
done

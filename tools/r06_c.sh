#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 400 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('phase', d['phase'], 'ms/step %.2f' % d['ms_per_step'], 'fwd %.2f bwd %.2f' % (d['forward_ms'], d['loss_backward_ms']))"

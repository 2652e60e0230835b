cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider -k "first_writer or deterministic_step or two_rank or two_fits or autograd or matches_oracle" 2>&1 | tail -3 > gpurun_out/r06_final4.log
timeout 300 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('phase', d['phase'], 'ms/step %.2f' % d['ms_per_step'], 'wgrad_stream', d['wgrad_stream'], d['wgrad_stream_timed_ms_on_off'])" >> gpurun_out/r06_final4.log
cat gpurun_out/r06_final4.log

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_ws3.log; : > $O
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider -k "first_writer or deterministic_step or winograd_domain or autograd" 2>&1 | tail -3 >> $O
for d in 0 1 auto; do
HVN_TRAIN_WGRAD_STREAM=$d timeout 400 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('HVN_TRAIN_WGRAD_STREAM=$d: phase', d['phase'], 'batch', d['batch'], 'ms/step %.2f' % d['ms_per_step'], 'fwd %.2f bwd %.2f' % (d['forward_ms'], d['loss_backward_ms']), 'wgrad_stream', d['wgrad_stream'], d['wgrad_stream_timed_ms_on_off'])" >> $O
done
cat $O

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_ws2.log; : > $O
for i in 1 2; do
timeout 400 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('auto: phase', d['phase'], 'batch', d['batch'], 'ms/step %.2f' % d['ms_per_step'], 'fwd %.2f bwd %.2f' % (d['forward_ms'], d['loss_backward_ms']), 'wgrad_stream', d['wgrad_stream'], d['wgrad_stream_timed_ms_on_off'], 'plan ops', d['plan_ops_per_step'])" >> $O
done

cat $O

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_tb.log; : > $O
P='import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d["variants"]["train_step"]
print(TAG, "value %.1f" % d["value"])
for ph in ("phase0", "phase1"):
    r = t[ph]; print(TAG, ph, "%.2f ms" % r["ms_per_step"], "fwd %.2f bwd %.2f" % (r["forward_ms"], r["loss_backward_ms"]), r["wgrad_stream"], r["wgrad_stream_timed_ms_on_off"], "atomic %.2f" % r.get("atomic_reduce_ms_per_step", 0))'
Q="--steps 5 --no-traffic --no-cfg3 --no-wsi-leg --no-roofline --no-cpu-baseline"
timeout 600 python bench.py $Q 2>/dev/null | python -c "TAG='shared streams, fitted, auto'
$P" >> $O
GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py $Q 2>/dev/null | python -c "TAG='shared streams, fitted, auto, GPU_MAX_HW_QUEUES=8'
$P" >> $O
cat $O

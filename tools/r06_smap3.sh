cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_sched_probe_cfg3.txt; : > $O
export PROBE_MODE=fast PROBE_TYPES=6 PROBE_BATCH=64 PROBE_DTYPE=bf16
for k in 0 5; do timeout 400 python tools/stream_map_probe.py $k 1,0 2,2 1,2 1,3 2,0 2>&1 | grep "idle\|Error" | tail -2 | sed 's/   pool offsets.*//' >> $O; done
cat $O

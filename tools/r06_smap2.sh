cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_sched_probe.txt; : > $O
for k in 0 5; do timeout 400 python tools/stream_map_probe.py $k 2,2 2,1 1,2 1,3 2,0 3,1 1,0 2>&1 | grep "idle\|Error" | tail -2 >> $O; done
cat $O

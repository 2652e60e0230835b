cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_stream_map_probe2.txt; : > $O
for k in 0 1 3; do timeout 200 python tools/stream_map_probe.py $k 2>&1 | grep "idle\|Error\|error" | tail -3 >> $O; done
HVN_STREAM_SELECT=0 timeout 200 python tools/stream_map_probe.py 3 2>&1 | grep "idle\|Error" | sed 's/^/HVN_STREAM_SELECT=0 /' >> $O
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_net.py -q -x -p no:cacheprovider 2>&1 | tail -3 >> $O
cat $O

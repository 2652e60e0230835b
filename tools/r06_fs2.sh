cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r06_fs2.log
cat gpurun_out/r06_fs2.log

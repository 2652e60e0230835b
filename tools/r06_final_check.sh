cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 -p no:cacheprovider 2>&1 | tail -24 > gpurun_out/r06_gpu_tests.log; tail -4 gpurun_out/r06_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

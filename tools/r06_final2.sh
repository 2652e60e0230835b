cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_two_ranks_one_gpu.py tests/test_gpu_trained_like.py tests/test_gpu_bf16.py -q -x -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/r06_final2_tests.log
HVN_KEEP_PMC_TABLE=gpurun_out/r06_traffic_by_kernel.txt timeout 900 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
cat gpurun_out/r06_final2_tests.log
python tools/bench_summary.py gpurun_out/r06_bench.json | head -24

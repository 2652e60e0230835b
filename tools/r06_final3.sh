cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -16 > gpurun_out/r06_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r06_smoke.log
HVN_KEEP_PMC_TABLE=gpurun_out/r06_traffic_by_kernel.txt timeout 900 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -4 gpurun_out/r06_gpu_tests.log; tail -2 gpurun_out/r06_smoke.log
python tools/bench_summary.py gpurun_out/r06_bench.json | head -16

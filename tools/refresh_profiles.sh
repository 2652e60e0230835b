cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/r_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r_smoke.log
python bench.py > gpurun_out/r_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r_prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r_prof.log 2>&1
export HVN_SPLIT=1 HVN_LANES=0
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r_pmcF -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/r_pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r_pmcW -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/r_pmcW.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d gpurun_out/r_pmcS -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/r_pmcS.log 2>&1
cat gpurun_out/r_gpu_tests.log gpurun_out/r_smoke.log; tail -1 gpurun_out/r_bench.log | cut -c1-200

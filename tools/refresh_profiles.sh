#!/bin/bash
# Round-end refresh on the GPU box: full -m gpu suite, smoke(), the default bench line, rocprofv3 kernel stats, the
# three PMC passes (separate runs: FETCH_SIZE, WRITE_SIZE, SQ), cfg-3 / fast-mode lines and the training bench.
# Outputs land in gpurun_out/; tools/{kernel_stats,pmc_traffic,pmc_sq,layer_table}.py turn them into profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r_smoke.log
python bench.py > gpurun_out/r_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r_prof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r_prof.log 2>&1
HVN_WINOGRAD3=64 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r_bench_w3_64.log 2>&1
python bench.py --dtype bf16 --mode fast --nr-types 6 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --quiet-net-output > gpurun_out/r_bench_cfg3_bf16.log 2>&1
python bench.py --dtype fp32 --mode fast --nr-types 6 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --quiet-net-output > gpurun_out/r_bench_cfg3_fp32.log 2>&1
python tools/train_bench.py --steps 8 --warmup 3 > gpurun_out/r_train_bench.jsonl 2> gpurun_out/r_train_bench.err
(
export HVN_SPLIT=1 HVN_LANES=0
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r_pmcF -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/r_pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r_pmcW -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/r_pmcW.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d gpurun_out/r_pmcS -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/r_pmcS.log 2>&1
)
cat gpurun_out/r_gpu_tests.log gpurun_out/r_smoke.log
for f in r_bench r_bench_w3_64 r_bench_cfg3_bf16 r_bench_cfg3_fp32; do tail -1 gpurun_out/$f.log | cut -c1-160; done
cat gpurun_out/r_train_bench.jsonl | cut -c1-200

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P -d gpurun_out/r06_pmcX$i -o p -- python bench.py --pmc-child > gpurun_out/r06_pmcX$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_stall.py gpurun_out/r06_pmc_stall.json $(find gpurun_out/r06_pmcX1 gpurun_out/r06_pmcX2 gpurun_out/r06_pmcX3 -name "*_results.db") 2>&1 | tail -12
tail -3 gpurun_out/r06_pmcX2.log
rm -rf gpurun_out/r06_pmcX1 gpurun_out/r06_pmcX2 gpurun_out/r06_pmcX3

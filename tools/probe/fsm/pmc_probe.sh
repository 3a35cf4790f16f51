# Build first: make -C tools/probe/fsm (the binary is git-ignored; it travels to the GPU box with the snapshot).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$i -- $R/tools/probe/fsm/fsm_probe $R/tests/golden/res/a_midsummer_nights_dream.txt 256 > $R/gpurun_out/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmc_$i.log; exit 1; }
  echo "pass $i ok"
done

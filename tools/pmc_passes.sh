cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-workload > $R/gpurun_out/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }
  echo "pass $tag ok"
done

#!/bin/bash
# On the GPU box: the PMC passes of tools/pmc_passes.sh over ONE extra workload (tools/run_workload.py), e.g. the 4 GiB uniform stream.
# Usage (through gpurun): bash tools/pmc_workload.sh <workload> [reps];  then here: python tools/pmc_summary.py gpurun_out/pmcw
set -e
R=$GRAFT_REPO_ROOT
w=$1; reps=${2:-2}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcw; mkdir -p $R/gpurun_out/pmcw
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcw/pmc_$tag -- python3 $R/tools/run_workload.py $w $reps > $R/gpurun_out/pmcw/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $R/gpurun_out/pmcw/$tag.log; exit 1; }
  echo "pass $tag ok"
done

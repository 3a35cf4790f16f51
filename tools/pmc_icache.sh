#!/bin/bash
# Instruction-cache counters of the bench's kernels (one rocprofv3 --pmc pass).  On the GPU box: bash tools/pmc_icache.sh [ET_LIB_PATH]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$1" ] && export ET_LIB_PATH=$1
rm -rf $R/gpurun_out/pmc_icache
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_icache -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-workload > $R/gpurun_out/pmc_icache.log 2>&1 || { echo "pass failed"; tail -5 $R/gpurun_out/pmc_icache.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k in ('k_tw_sync','k_dec_write_wave','k_encode_tiles','k_hist_tiles'):
    v=d.get(k,{})
    print(k, {c:round(x) for c,x in v.items() if c.startswith('SQC') or c in ('SQ_IFETCH','SQ_INSTS_VALU','SQ_WAIT_INST_ANY')})
"

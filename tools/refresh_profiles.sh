#!/bin/bash
# On the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes.
# Usage (from the repo root, through gpurun): bash tools/refresh_profiles.sh <tag>
# Afterwards, here: python tools/collect_profiles.py <tag>
set -e
R=$GRAFT_REPO_ROOT
tag=$1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_* $R/gpurun_out/prof_$tag  # (clear the LOCAL gpurun_out/pmc_* too before calling gpurun: merged files accumulate)
timeout -k 10 400 python3 $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
echo "bench ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-cpu-baseline --no-second-workload > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_rocprof.err
echo "rocprof ok"
bash $R/tools/pmc_passes.sh

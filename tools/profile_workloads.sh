#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of bench.py's extra workloads, each on its own (tools/run_workload.py).
# Usage (through gpurun, from the repo root): bash tools/profile_workloads.sh <tag> [workload ...]
set -e
R=$GRAFT_REPO_ROOT
tag=$1; shift
names=${@:-uniform255-4G text-100M text-5M uniform256-16G}
cd /tmp && export TMPDIR=/tmp
for w in $names; do
  rm -rf $R/gpurun_out/prof_${tag}_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$w -- python3 $R/tools/run_workload.py $w > $R/gpurun_out/${tag}_$w.json 2> $R/gpurun_out/${tag}_$w.err || { echo "$w FAILED"; tail -5 $R/gpurun_out/${tag}_$w.err; exit 1; }
  echo "$w ok: $(cat $R/gpurun_out/${tag}_$w.json | cut -c1-400)"
done

#!/usr/bin/env python3
"""One of bench.py's extra workloads on its own (what rocprofv3 is pointed at for that stream's kernel stats and PMC passes):
    python3 tools/run_workload.py uniform255-4G [reps]        -> one JSON line (bench.py's "workloads" entry for it)
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_u255 -- python3 tools/run_workload.py uniform255-4G 5
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch

    import bench
    import entreepy_amd as E
    from entreepy_amd import sharded
    from tests import corpus

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = E.Context(0)
    ctx.use_torch_stream()
    ctx.enable_timing(True)
    pipe = sharded.ShardedCodec(ctx, None, dev)
    out = bench.run_extra_workload(name, torch, E, corpus, ctx, pipe, dev, 1 << 30, os.environ.get("ET_BENCH_NO_VERIFY") != "1", reps)
    os.write(real_stdout, (json.dumps({name: out}) + "\n").encode())


if __name__ == "__main__":
    main()

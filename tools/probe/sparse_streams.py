#!/usr/bin/env python3
"""Round trip of streams with one dominant byte value (a share p of zeros, the rest uniform over 1..254): codes of 1-2 bits for the
zero, so a 256-bit subsequence holds up to ~200 symbols -- more than a wavefront's stage of the write pass takes at once (windows).
    python3 tools/probe/sparse_streams.py [bytes] [p,p,...]      -> one JSON line per p"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
    ps = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.5, 0.75, 0.9, 0.97, 0.99]
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch

    import bench
    import entreepy_amd as E
    from entreepy_amd import sharded

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = E.Context(0)
    ctx.use_torch_stream()
    pipe = sharded.ShardedCodec(ctx, None, dev)
    enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device=dev)
    dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    for p in ps:
        g.manual_seed(int(p * 1000))
        text = bench.uniform_bytes_torch(n, 1, 255, 77, dev)  # (254 other values: with all 256 present the reference drops one)
        step = 1 << 26
        for s in range(0, n, step):
            m = min(step, n - s)
            text[s : s + m][torch.rand(m, generator=g, device=dev) < p] = 0
        out = bench.measure_stream(torch, ctx, pipe, text, enc, dec, 5, 3, True, os.environ.get("ET_BENCH_NO_VERIFY") != "1")  # (probe builds give wrong output on purpose)
        line = {"p_zero": p, "code_lengths": out["code_lengths"], "packed_bytes": out["packed_bytes"], "ms_per_step": out["ms_per_step"], "round_trip_GBps": out["round_trip_GBps"],
                "encode_GBps": out["encode_GBps"], "decode_GBps": out["decode_GBps"], "phase_ms": out["phase_ms"], "decode_path": out["decode_path"], "verified": out["verified"]}
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
        del text


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Round trip of uniform draws over k byte values, k = 2 .. 255 (bench.measure_stream; 256 MiB each): which decode path such a
stream takes and what it costs.  Flat alphabets are where the reference's own README has its worst case (BASELINE configs[4]);
k = 4 is what a DNA sequence's code looks like (four 2-bit codewords), k = 16 a hex dump's, k = 64 base64's.

    python3 tools/probe/flat_alphabets.py [bytes] [k,k,...]      -> one JSON line per k
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
    ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4, 10, 16, 26, 64, 100, 128, 200, 255]
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
    for k in ks:
        text = bench.uniform_bytes_torch(n, 1, 1 + k, 0x5EED0F00 + k, dev)
        out = bench.measure_stream(torch, ctx, pipe, text, enc, dec, 5, 3)
        line = {"k": k, "code_lengths": out["code_lengths"], "ms_per_step": out["ms_per_step"], "round_trip_GBps": out["round_trip_GBps"],
                "encode_GBps": out["encode_GBps"], "decode_GBps": out["decode_GBps"], "dec_sync_ms": out["phase_ms"]["dec_sync"],
                "dec_body_ms": out["phase_ms"]["dec_body"], "decode_path": out["decode_path"], "verified": out["verified"]}
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
        del text


if __name__ == "__main__":
    main()

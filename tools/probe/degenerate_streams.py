#!/usr/bin/env python3
"""Round trip of 256 MiB streams of one, two and three byte values (bench.measure_stream): a lone symbol encodes to the bare header
(the reference's Q2: no inverse -- bench.fail says so and the line shows "exit"), two values are two 1-bit codewords (k_fixed_write),
three with two of them rare put ~250 symbols into a 256-bit subsequence (the chained write's windows: DESIGN section 7, 3b).
    python3 tools/probe/degenerate_streams.py      -> one JSON line per case"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
import entreepy_amd as E
from entreepy_amd import sharded
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = E.Context(0); ctx.use_torch_stream()
pipe = sharded.ShardedCodec(ctx, None, dev)
n = 1 << 28
enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device=dev)
dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
cases = {}
cases["one value"] = torch.full((n,), 65, dtype=torch.uint8, device=dev)
t = torch.full((n,), 65, dtype=torch.uint8, device=dev); t[torch.rand(n, generator=g, device=dev) < 0.001] = 66
cases["two values, 0.1 % of the second"] = t
t = torch.full((n,), 65, dtype=torch.uint8, device=dev); t[::4096] = 66; t[1::4096] = 67
cases["three values, two of them rare"] = t
for name, text in cases.items():
    try:
        out = bench.measure_stream(torch, ctx, pipe, text, enc, dec, 5, 3)
        print(json.dumps({"case": name, "code_lengths": out["code_lengths"], "symbols": out["symbols"], "packed": out["packed_bytes"], "ms": out["ms_per_step"], "enc": out["encode_GBps"], "dec": out["decode_GBps"], "phase": out["phase_ms"], "path": out["decode_path"], "ok": out["verified"]}))
    except SystemExit as e:
        print(json.dumps({"case": name, "exit": str(e)}))
    except Exception as e:
        print(json.dumps({"case": name, "error": repr(e)}))

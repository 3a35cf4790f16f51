#!/usr/bin/env python3
"""Wall-clock of the two library calls against the sum of their GPU phases (where does the
host leave the GPU idle?).  python tools/phase_gaps.py [bytes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entreepy_amd as E
from entreepy_amd import sharded
from tests import corpus

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
dev = torch.device("cuda", 0)
text = corpus.text_like_torch(n, 0x5EED0004, dev)
ctx = E.Context(0); ctx.use_torch_stream(); ctx.reserve(n); ctx.enable_timing(True)
enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device=dev)
dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
pipe = sharded.ShardedCodec(ctx, None, dev)
for timing in (True, False):
    ctx.enable_timing(timing)
    for _ in range(3):
        r = pipe.encode_shard(text, enc); pipe.decode_shard(enc, r, dec)
    torch.cuda.synchronize()
    te = td = 0.0; K = 10
    for _ in range(K):
        t0 = time.perf_counter(); r = pipe.encode_shard(text, enc); torch.cuda.synchronize(); t1 = time.perf_counter()
        pipe.decode_shard(enc, r, dec); torch.cuda.synchronize(); t2 = time.perf_counter()
        te += t1 - t0; td += t2 - t1
    print(f"timing={timing}: encode wall {te/K*1e3:.3f} ms, decode wall {td/K*1e3:.3f} ms")
    if timing:
        print("  encode phases", {k: round(v, 4) for k, v in r["timings"].items()})
        print("  decode phases", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ctx.timings().items()})

#!/usr/bin/env python3
"""K4's two "process modes" (0.342 / 0.354 ms, fixed for a process's lifetime, "follow physical placement"), run to
ground: text, image and output cut from ONE slab at chosen relative offsets, K4 timed per placement inside
one process -- does its time follow the RELATIVE placement of its buffers (steerable: pick the fast one), or only the process
(the driver's virtual-to-physical mapping: not steerable from here)?
    python tools/probe/k4_placement.py [steps]          (run it in several processes: the script prints its pid and the slab's address)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import entreepy_amd as E
from entreepy_amd import sharded
from tests import corpus

n = 1 << 30
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
src = corpus.text_like_torch(n, 0x5EED0004, dev)
bound = E.encode_bound(n) + 64
slab = torch.empty(4 * n + (64 << 20), dtype=torch.uint8, device=dev)
base = slab.data_ptr()
pad = (-base) % (2 << 20)  # the slab's first 2 MiB boundary
ctx = E.Context(0)
ctx.reserve(n)
ctx.enable_timing(True)
pipe = sharded.ShardedCodec(ctx, None, dev)
print(f"pid {os.getpid()} slab at {base:#x} (+{pad} to 2 MiB)", flush=True)
print("text_off   image_off(rel. to text end, 2 MiB-aligned +)   hist    K4      D1      D3     [ms]", flush=True)
M = 1 << 20
for t_off, e_off in ((0, 0), (0, 4096), (0, 64 << 10), (0, 1 * M), (0, 3 * M), (0, 17 * M), (4096, 0), (1 * M, 0), (1 * M, 1 * M), (0, 0)):
    a0 = pad + t_off
    text = slab[a0 : a0 + n]
    text.copy_(src)
    b0 = a0 + n
    b0 += (-(base + b0)) % (2 * M) + e_off
    enc = slab[b0 : b0 + bound]
    c0 = b0 + bound
    c0 += (-(base + c0)) % (2 * M)
    dec = slab[c0 : c0 + n + 64]
    acc = {"hist": 0.0, "enc_body": 0.0, "sync": 0.0, "body": 0.0}
    for i in range(steps + 10):
        r = pipe.encode_shard(text, enc, timings=False)
        m = pipe.decode_shard(enc, r, dec)
        te, td = pipe.encode_timings(), ctx.timings("decode")
        if i >= 10:
            acc["hist"] += te["hist"]
            acc["enc_body"] += te["enc_body"]
            acc["sync"] += td["sync_first_ms"]
            acc["body"] += td["body_ms"]
    torch.cuda.synchronize()
    assert m == n and torch.equal(dec[:n], text)
    print(f"{t_off:8d}   {e_off:10d}                                   {acc['hist'] / steps:.4f}  {acc['enc_body'] / steps:.4f}  {acc['sync'] / steps:.4f}  {acc['body'] / steps:.4f}", flush=True)

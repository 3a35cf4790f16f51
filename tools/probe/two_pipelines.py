"""Experiment: S independent encode+decode pipelines (own context, own stream, own host thread) on one GPU."""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import entreepy_amd as E
from entreepy_amd import sharded
from tests import corpus

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << 30
dev = torch.device("cuda:0")
text = corpus.text_like_torch(n, 0x5EED0004, dev)
pipes = []
for s in range(S):
    ctx = E.Context(0)
    ctx.reserve(n)
    enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device=dev)
    dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    pipe = sharded.ShardedCodec(ctx, None, dev)
    ctx.use_own_stream()
    pipes.append((ctx, pipe, enc, dec))

def run(p, k):
    ctx, pipe, enc, dec = p
    for _ in range(k):
        r = pipe.encode_shard(text, enc, timings=False)
        m = pipe.decode_shard(enc, r, dec)
    assert m == n

for p in pipes:
    run(p, 2)
torch.cuda.synchronize()
for p in pipes:
    assert torch.equal(p[3][:n], text)
per = K // S
t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(p, per)) for p in pipes]
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
el = time.perf_counter() - t0
for p in pipes:
    assert torch.equal(p[3][:n], text)
print(f"streams {S}: {per * S} steps in {el * 1e3:.2f} ms -> {per * S * n / el / 1e9:.1f} GB/s, {el / (per * S) * 1e3:.4f} ms per step")

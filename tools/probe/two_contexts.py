#!/usr/bin/env python3
"""Do two contexts on ONE GPU, each on its own stream and host thread, move more text per second than one?

The step's four large kernels lean on different parts of the chip (PMC, DESIGN section 6: K1 the memory, K4 the VALU, D1 and D3
the LDS pipe), and a step hands over to the host twice (~20 us each with the GPU idle).  Two independent objects coded side by
side -- what a service with more than one request in flight does -- fill those holes.  This is NOT bench.py's `value` (one
context, one step after the other); it is printed beside it.

    python3 tools/probe/two_contexts.py [bytes] [steps]
"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    import torch

    import entreepy_amd as E
    from entreepy_amd import sharded
    from tests import corpus

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_ctx = int(os.environ.get("ET_PROBE_CONTEXTS", 2))
    if os.environ.get("ET_PROBE_DATA") == "uniform255":  # (the row walk's kernels: k_row_sync the VALU, k_row_write the memory)
        import bench

        texts = [bench.uniform_bytes_torch(n, 1, 256, 0x5EED0255 + i, dev) for i in range(n_ctx)]
    else:
        texts = [corpus.text_like_torch(n, 0x5EED0004 + i, dev) for i in range(n_ctx)]
    decode_only = os.environ.get("ET_PROBE_DECODE_ONLY") == "1"
    lanes = []
    for i in range(n_ctx):
        ctx = E.Context(0)
        ctx.use_own_stream()
        ctx.reserve(n)
        ctx.enable_timing(False)
        enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device=dev)
        dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        lanes.append((ctx, sharded.ShardedCodec(ctx, None, dev), texts[i], enc, dec))
    torch.cuda.synchronize()

    def run(lane, k):
        ctx, pipe, text, enc, dec = lane
        r = pipe.encode_shard(text, enc, timings=False)
        for _ in range(k):
            if not decode_only:
                r = pipe.encode_shard(text, enc, timings=False)
            pipe.decode_shard(enc, r, dec)

    def timed(active, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if len(active) == 1:
            run(active[0], k)
        else:
            th = [threading.Thread(target=run, args=(lane, k)) for lane in active]
            for t in th:
                t.start()
            for t in th:
                t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for lane in lanes:  # warm-up + check
        run(lane, 20)
        torch.cuda.synchronize()
        assert torch.equal(lane[4][:n], lane[2]), "round trip differs"
    out = {"bytes": n, "steps": steps}
    for rep in range(2):
        t1 = timed(lanes[:1], steps)
        tn = timed(lanes, steps)
        out[f"one_context_GBps_{rep}"] = round(n * steps / t1 / 1e9, 1)
        out[f"{n_ctx}_contexts_GBps_{rep}"] = round(n_ctx * n * steps / tn / 1e9, 1)
    for lane in lanes:
        assert torch.equal(lane[4][:n], lane[2]), "round trip differs"
    print(json.dumps(out))


if __name__ == "__main__":
    main()

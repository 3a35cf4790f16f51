"""Soak of the encode path against the oracle: random sources (text, skewed, uniform alphabets,
geometric -> long codes, Fibonacci-like counts -> codes beyond 32 bits), random sizes, random
tile geometry (et_ctx_set_tile_rounds), device entry with a misaligned input pointer, and the
host entry; every .et image must equal oracle.encode's.  Also round-trips through the decoder.
Usage: python tests/soak/soak_encode.py SEED TRIALS [MAX_BYTES]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import entreepy_amd as E
from oracle import oracle as O
from tests import corpus


def main():
    seed, trials = int(sys.argv[1]), int(sys.argv[2])
    max_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 3_000_000
    rng = np.random.default_rng(seed)
    ctx = E.Context(0)
    ctx.use_torch_stream()
    t0 = time.time()
    bad = 0
    for trial in range(trials):
        n = int(rng.integers(1, max_bytes)) if rng.random() < 0.8 else int(rng.integers(1, 5000))
        src = int(rng.integers(0, 5))
        if src == 0:
            text = corpus.text_like(n, seed * 100_000 + trial)
        elif src == 1:
            text = corpus.uniform(n, seed * 100_000 + trial, 0, int(rng.integers(1, 257)))
        elif src == 2:
            p = float(rng.choice([0.5, 0.9, 0.999]))
            text = np.where(rng.random(n) < p, int(rng.integers(0, 256)), corpus.text_like(n, seed * 100_000 + trial)).astype(np.uint8)
        elif src == 3:
            text = np.minimum(rng.geometric(0.5, size=n) - 1, int(rng.integers(8, 40))).astype(np.uint8)
        else:  # Fibonacci-like counts: a degenerate tree, codes of up to ~40 bits (quirk Q3 territory)
            k = int(rng.integers(20, 42))
            fib = [1, 1]
            while len(fib) < k:
                fib.append(fib[-1] + fib[-2])
            reps = np.array(fib, dtype=np.float64)
            reps = np.maximum(1, (reps * min(1.0, 2_000_000 / reps.sum())).astype(np.int64))
            text = rng.permutation(np.repeat(np.arange(k, dtype=np.uint8), reps))
            n = text.size
        want = O.encode(text)
        ctx.set_tile_rounds(int(rng.choice([0, 0, 1, 2, 4, 8, 16])))
        mode = int(rng.integers(0, 3))
        if mode == 2 and int(O.build_dict(O.histogram(text))[1].max()) > 32:
            mode = 0  # (the staged calls take code tables of up to 32 bits)
        if mode == 0:
            got = ctx.encode(text)
        elif mode == 2:  # virtual shards: uneven (also empty / tiny) pieces, one table, bit-offset concat
            shards = int(rng.integers(2, 9))
            bounds = np.sort(np.concatenate([[0, n], rng.integers(0, n + 1, size=shards - 1)])).astype(np.int64)
            t = torch.from_numpy(np.ascontiguousarray(text)).cuda()
            hists = []
            for r in range(shards):
                h = torch.zeros(256, dtype=torch.int64, device="cuda")
                if bounds[r + 1] > bounds[r]:
                    ctx.histogram_device(t[bounds[r] : bounds[r + 1]], h)
                hists.append(h.cpu().numpy().astype(np.uint64))
            cb = E.Codebook.from_histogram(np.sum(hists, axis=0).astype(np.uint64))
            header = cb.header(n)
            img = np.zeros(len(want) + 16, dtype=np.uint8)
            img[: len(header)] = np.frombuffer(header, dtype=np.uint8)
            bit = len(header) * 8
            for r in range(shards):
                view = t[bounds[r] : bounds[r + 1]]
                if view.numel() == 0:
                    continue
                ctx.histogram_device(view, torch.zeros(256, dtype=torch.int64, device="cuda"))  # the staged protocol: K1 first
                out = torch.zeros(view.numel() * 4 + 64, dtype=torch.uint8, device="cuda")
                end = ctx.encode_body_device(cb, view, out, bit % 32)
                torch.cuda.synchronize()
                piece = out[: (end + 7) // 8].cpu().numpy()
                base_byte = (bit // 32) * 4
                img[base_byte : base_byte + piece.size] |= piece
                bit += end - bit % 32
            got = img[: (bit + 7) // 8].tobytes()
        else:
            lo = int(rng.integers(0, 16))
            buf = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
            buf[lo : lo + n] = torch.from_numpy(np.ascontiguousarray(text)).cuda()
            out = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device="cuda")
            m = ctx.encode_device(buf[lo : lo + n], out)
            torch.cuda.synchronize()
            got = out[:m].cpu().numpy().tobytes()
        ok = got == want
        if ok and len(set(text.tolist()[:100000])) > 1:
            try:
                ok = ctx.decode(got[4:]) == O.decode(want[4:])
            except E.EntreepyError as e:  # codes beyond 32 bits: the decoder declines, as documented
                ok = "32" in str(e) or "unsupported" in str(e).lower()
        if not ok:
            bad += 1
            print("trial", trial, "src", src, "n", n, "mode", mode, "MISMATCH", len(got), len(want), flush=True)
        if trial % 50 == 0:
            print(f"trial {trial} ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"done: {trials} trials, bad = {bad}, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

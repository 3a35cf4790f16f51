"""Long fuzz soak of the decode path on the GPU box (tests/test_gpu_parity.py holds the short,
fixed-seed versions): corrupted bodies must decode to exactly what the oracle says, corrupted
dictionaries must return an error or bounded output, and nothing may hang -- run it under
`timeout`, progress goes to stdout once per 50 trials.
Usage: python tests/soak/soak_fuzz.py SEED TRIALS [MAX_BYTES]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

import entreepy_amd as E
from oracle import oracle as O
from tests import corpus


def main():
    seed, trials = int(sys.argv[1]), int(sys.argv[2])
    max_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 700_000
    ctx = E.Context(0)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = errors = 0
    for trial in range(trials):
        if trial % 97 == 0:
            ctx.enable_timing(trial % 2 == 0)  # both launch flavours: with and without kernel-carried events
        n = int(rng.integers(20_000, max_bytes))
        src = int(rng.integers(0, 6))
        if src == 4:  # flat alphabets of 128 .. 255 symbols: complete codes of 7 and 8 bits -- the row walk (csrc/et_rowsync.hip)
            k = int(rng.integers(128, 256)) if rng.integers(0, 8) else 128
            if rng.integers(0, 4) == 0:  # ... and fixed-length codes: 4 .. 64 symbols (k_fixed_write)
                k = 1 << int(rng.integers(2, 7))
            vals = (np.arange(k) + int(rng.integers(0, 257 - k))).astype(np.uint8)
            text = np.tile(vals, n // k + 1)[:n]
            rng.shuffle(text)
        elif src == 5:  # one dominant value (1- or 2-bit codeword) among 2 .. 254 others: > 128 symbols per subsequence -- the write pass's strips
            text = rng.integers(1, 2 + int(rng.integers(2, 254)), size=n).astype(np.uint8)
            text[rng.random(n) < float(rng.choice([0.9, 0.95, 0.99, 0.999]))] = 0
        elif src == 0:
            text = corpus.text_like(n, seed * 100_000 + trial)
        elif src == 1:
            text = corpus.uniform(n, seed * 100_000 + trial, 1, 1 + int(rng.integers(2, 255)))
        elif src == 2:
            p = float(rng.choice([0.5, 0.9, 0.99, 0.999]))
            text = np.where(rng.random(n) < p, int(rng.integers(0, 256)), corpus.text_like(n, seed * 100_000 + trial)).astype(np.uint8)
        else:  # geometric lengths: long codes (up to ~30 bits)
            k = int(rng.integers(8, 30))
            text = np.minimum(rng.geometric(0.5, size=n) - 1, k).astype(np.uint8)
        good = bytearray(O.encode(text)[4:])
        try:
            _, _, off = E.parse_header(bytes(good))
        except E.EntreepyError:
            continue
        if len(good) - off < 64:
            continue
        header_fuzz = rng.random() < 0.2
        for _ in range(int(rng.integers(1, 6))):
            if header_fuzz:
                good[int(rng.integers(0, off))] ^= 1 << int(rng.integers(0, 8))
                continue
            k, a, ln = int(rng.integers(0, 5)), int(rng.integers(off, len(good))), int(rng.integers(1, 40_000))
            if k == 0:
                good[a] ^= 1 << int(rng.integers(0, 8))
            elif k == 1:
                good[a : a + ln] = rng.integers(0, 256, size=len(good[a : a + ln]), dtype=np.uint8).tobytes()
            elif k == 2:
                good[a : a + ln] = b"\xff" * len(good[a : a + ln])
            elif k == 3:
                good[a : a + ln] = b"\x00" * len(good[a : a + ln])
            else:
                del good[max(off + 1, a) :]
        data = bytes(good)
        try:
            got = ctx.decode(data)
        except E.EntreepyError:
            errors += 1
            continue
        if header_fuzz:
            _, declared, _ = E.parse_header(data)
            if len(got) > declared:
                bad += 1
                print("trial", trial, "UNBOUNDED output", len(got), declared, flush=True)
            continue
        want = O.decode(data)
        if got != want:
            bad += 1
            print("trial", trial, "src", src, "MISMATCH", len(got), len(want), flush=True)
        if trial % 50 == 0:
            print(f"trial {trial} ok ({time.time() - t0:.0f} s, {errors} rejected)", flush=True)
    print(f"done: {trials} trials, bad = {bad}, rejected = {errors}, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

"""Soak of the decode's choice of path on (nearly) flat alphabets, on the GPU box: k = 2 .. 255 symbols with weights that are equal
or nearly so (every code of one or two neighbouring lengths: fixed-length -> k_fixed_write, L and L + 1 bits -> the tree walk when
et::quick_to_synchronise says so, else the exit maps; 7 and 8 bits -> the row walk), streams of 1 KB .. MAX_BYTES, encoded and
decoded on the device and compared there; every 4th trial also decodes a truncated copy and holds it against the oracle (small
streams only).  Whatever the first sweep makes of a stream that does not settle, the result must be the text.
Usage: python tests/soak/soak_flat.py SEED TRIALS [MAX_BYTES [MIN_BYTES]]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import entreepy_amd as E
from oracle import oracle as O


def main():
    seed, trials = int(sys.argv[1]), int(sys.argv[2])
    max_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 24_000_000
    min_bytes = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
    ctx = E.Context(0)
    ctx.use_torch_stream()
    ctx.enable_timing(True)
    rng = np.random.default_rng(seed)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    t0 = time.time()
    bad = 0
    paths = {"fixed": 0, "row": 0, "tree walk": 0, "exit maps": 0, "tree walk then exit maps": 0}
    enc = torch.zeros(E.encode_bound(max_bytes) + 64, dtype=torch.uint8, device="cuda")
    dec = torch.empty(max_bytes + 64, dtype=torch.uint8, device="cuda")
    for trial in range(trials):
        k = int(rng.integers(2, 256))
        n = int(10 ** rng.uniform(np.log10(min_bytes), np.log10(max_bytes)))
        lo = int(rng.integers(0, 257 - k))
        if rng.integers(0, 3) == 0:  # weights within a factor of ~1.5 of each other: still one or two neighbouring lengths, other codes
            w = torch.from_numpy(rng.uniform(1.0, 1.5, size=k)).cuda()
            text = (torch.multinomial(w, n, replacement=True, generator=g) + lo).to(torch.uint8)
        else:
            text = (torch.randint(0, k, (n,), generator=g, device="cuda", dtype=torch.int16) + lo).to(torch.uint8)
        ln = ctx.encode_device(text, enc)
        m = ctx.decode_device(enc[4:ln], dec)
        t = ctx.timings("decode")
        ok = m == n and bool(torch.equal(dec[:n], text))
        path = ("fixed" if t["fixed_sync"] else "row" if t["row_sync"] else
                ("tree walk then exit maps" if t["sync_first_ms"] > 0 else "exit maps") if t["exhaustive_sync"] else "tree walk")
        paths[path] += 1
        if ok and trial % 4 == 0 and n < 300_000:
            et = enc[:ln].cpu().numpy().tobytes()
            cut = int(rng.integers(1, min(400, len(et) - 8)))
            ok = ctx.decode(et[4:-cut]) == O.decode(et[4:-cut])
        if not ok:
            bad += 1
            print(f"BAD trial {trial}: k {k} n {n} lo {lo} path {path}", flush=True)
        if trial % 50 == 0:
            print(f"trial {trial} ok ({time.time() - t0:.0f} s) {paths}", flush=True)
    print(f"done: {trials} trials, bad = {bad}, paths {paths}, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

"""Soak of the cold multi-GPU decode entry points (et_decode_range_sync / _write) with
virtual ranks on one GPU: random sources, sizes, rank counts and UNEVEN block ranges, clean
and truncated streams; the exchange is done by hand as sharded.decode_cold does it.  The
concatenated pieces must equal the oracle's decode.  Run under `timeout`.
Usage: python tests/soak/soak_cold.py SEED TRIALS"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import entreepy_amd as E
from oracle import oracle as O
from entreepy_amd.sharded import cut_blocks
from tests import corpus


def main():
    seed, trials = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    ctxs = [E.Context(0) for _ in range(6)]
    for c in ctxs:
        c.use_torch_stream()
    t0 = time.time()
    bad = 0
    for trial in range(trials):
        n = int(rng.integers(60_000, 3_000_000))
        src = int(rng.integers(0, 3))
        if src == 0:
            text = corpus.text_like(n, seed * 100_000 + trial)
        elif src == 1:
            p = float(rng.choice([0.5, 0.9, 0.99]))
            text = np.where(rng.random(n) < p, int(rng.integers(0, 256)), corpus.text_like(n, seed * 100_000 + trial)).astype(np.uint8)
        else:
            text = np.minimum(rng.geometric(0.5, size=n) - 1, int(rng.integers(8, 30))).astype(np.uint8)
        et = O.encode(text)[4:]
        cb, n_symbols, body_off = E.parse_header(et)
        if int(np.asarray(cb.length).max()) > 32:
            continue
        if rng.random() < 0.3:  # truncated stream
            et = et[: int(rng.integers(body_off + 40_000, len(et)))] if len(et) > body_off + 40_001 else et
        want = O.decode(et)
        comp = torch.from_numpy(np.frombuffer(et, dtype=np.uint8).copy()).cuda()
        ptr = comp.data_ptr() + body_off
        base_off, first_bit = body_off - (ptr & 3), (ptr & 3) * 8
        stream = comp[base_off:]
        n_blocks = cut_blocks(stream.numel())
        ranks = int(rng.integers(2, 7))
        cuts = sorted(set([0, n_blocks] + [int(x) for x in rng.integers(1, max(2, n_blocks), size=ranks - 1)]))
        infos, spans, used = [], [], []
        for r in range(len(cuts) - 1):
            lo, hi = cuts[r], cuts[r + 1]
            begin, end = lo * 8192, (stream.numel() if hi == n_blocks else hi * 8192)
            c = ctxs[r]
            infos.append(c.decode_range_sync(cb, stream, begin, end, first_bit if lo == 0 else -1))
            spans.append((begin, end))
            used.append(c)
        settled = False
        for _ in range(len(used) + 3):
            prev, wrong = first_bit, []
            for i, inf in enumerate(infos):
                if inf["start_bit"] != prev:
                    wrong.append((i, prev))
                prev = inf["exit_bit"]
            if not wrong:
                settled = True
                break
            for i, w in wrong:
                infos[i] = used[i].decode_range_sync(cb, stream, spans[i][0], spans[i][1], w)
        out, first = [], 0
        for c, inf in zip(used, infos):
            take = max(0, min(inf["n_symbols"], n_symbols - first))
            buf = torch.empty(inf["n_symbols"] + 64, dtype=torch.uint8, device="cuda")
            m = c.decode_range_write(take, buf)
            torch.cuda.synchronize()
            out.append(buf[:m].cpu().numpy())
            first += inf["n_symbols"]
        got = np.concatenate(out).tobytes()
        if not settled or got != want:
            bad += 1
            print("trial", trial, "src", src, "ranks", len(used), "cuts", cuts, "settled", settled, "MISMATCH", len(got), len(want), flush=True)
        if trial % 50 == 0:
            print(f"trial {trial} ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"done: {trials} trials, bad = {bad}, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

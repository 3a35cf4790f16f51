"""Soak of the group calls behind the C ABI (csrc/et_shard_seq.cpp on the GPU backend): rank threads on one GPU, an
in-memory exchange; random sources, sizes, world sizes and RAGGED cuts (empty shards, shards of a few bytes), clean
images from dirty buffers; et_encode_sharded -> et_shard_merge_seams -> et_shard_place must equal the oracle's image,
et_decode_sharded (whole stream on every rank) and window -> begin -> write (a rank's own bytes only) must return the
oracle's decode.  Every few trials one rank is made to fail (a missing or too small buffer): all ranks must return
that status and the groups must work again afterwards.  Run under `timeout`.
Usage: python tests/soak/soak_sharded.py SEED TRIALS"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import entreepy_amd as E
from entreepy_amd.codec import Group
from oracle import oracle as O
from tests import corpus


class ThreadGather:
    def __init__(self, world):
        self.world, self.slots, self.bar = world, [None] * world, threading.Barrier(world)

    def of(self, rank):
        def gather(mine):
            self.slots[rank] = mine
            self.bar.wait(timeout=60)
            out = b"".join(self.slots)
            self.bar.wait(timeout=60)
            return out

        return gather


def run_ranks(world, body):
    out, threads = [None] * world, []

    def wrap(r):
        try:
            out[r] = ("ok", body(r))
        except BaseException as e:  # noqa: BLE001
            out[r] = ("err", e)

    for r in range(world):
        threads.append(threading.Thread(target=wrap, args=(r,)))
        threads[-1].start()
    for t in threads:
        t.join(timeout=180)
    if any(t.is_alive() for t in threads):
        raise SystemExit("a rank hung")
    return out


def main():
    seed, trials = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    t0 = time.time()
    bad = enc_ok = dec_ok = win_ok = fail_ok = 0
    for trial in range(trials):
        world = int(rng.integers(1, 6))
        n = int(rng.integers(1, 2_500_000)) if rng.random() < 0.9 else int(rng.integers(1, 200))
        src = int(rng.integers(0, 4))
        if src == 0:
            text = corpus.text_like(n, seed * 100_000 + trial)
        elif src == 1:
            p = float(rng.choice([0.5, 0.9, 0.99]))
            text = np.where(rng.random(n) < p, int(rng.integers(0, 256)), corpus.text_like(n, seed * 100_000 + trial)).astype(np.uint8)
        elif src == 2:
            text = corpus.uniform(n, seed * 100_000 + trial, 0, int(rng.choice([2, 3, 17, 200, 255, 256])))
        else:
            text = np.minimum(rng.geometric(0.5, size=n) - 1, int(rng.integers(8, 30))).astype(np.uint8)
        cuts = sorted([0, n] + [int(x) for x in rng.integers(0, n + 1, size=world - 1)])
        if rng.random() < 0.3 and world > 2:
            cuts[2] = cuts[1]  # an empty shard
        want = O.encode(text)
        texts = [torch.from_numpy(text[cuts[r] : cuts[r + 1]].copy()).to(dev) for r in range(world)]
        encs = [torch.full((E.encode_bound(t.numel()) + 64,), 0xA5, dtype=torch.uint8, device=dev) for t in texts]
        image = torch.full(((len(want) + 3) // 4 * 4,), 0xEE, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        x = ThreadGather(world)
        ctxs = [E.Context(0) for _ in range(world)]
        groups = [Group(ctxs[r], r, world, allgather=x.of(r)) for r in range(world)]
        try:
            if trial % 4 == 3 and world > 1:  # one rank fails first: everybody must hear of it
                victim = int(rng.integers(0, world))
                res = run_ranks(world, lambda r: groups[r].encode_sharded(texts[r], None if r == victim else encs[r]))
                if not all(k == "err" and getattr(e, "status", None) == 6 for k, e in res):
                    bad += 1
                    print("FAIL (failure protocol)", seed, trial, res, flush=True)
                else:
                    fail_ok += 1

            def enc_rank(r):
                groups[r].encode_sharded(texts[r], encs[r])
                groups[r].merge_seams(encs[r])
                groups[r].place(encs[r], image)
                torch.cuda.synchronize()

            res = run_ranks(world, enc_rank)
            # a shard of symbols that are rare in the whole text may not fit et_encode_bound(n): then ALL ranks say so
            if any(k == "err" for k, _ in res):
                if not all(k == "err" and getattr(e, "status", None) == 3 for k, e in res):
                    bad += 1
                    print("FAIL (encode)", seed, trial, world, n, src, cuts, res, flush=True)
                continue
            if image[: len(want)].cpu().numpy().tobytes() != want:
                bad += 1
                print("FAIL (image)", seed, trial, world, n, src, cuts, flush=True)
                continue
            enc_ok += 1
            cb, n_symbols, body_off = E.parse_header(want[4:])
            if int(np.asarray(cb.length).max()) > 32:
                continue
            stream = want[4:]
            if rng.random() < 0.3 and len(stream) > body_off + 20_000:  # a truncated stream: the oracle's decode of what is left
                stream = stream[: int(rng.integers(body_off + 10_000, len(stream)))]
            truth = O.decode(stream)
            comp = torch.frombuffer(bytearray(stream), dtype=torch.uint8).to(dev)
            outs = [torch.zeros(n + 64, dtype=torch.uint8, device=dev) for _ in range(world)]
            torch.cuda.synchronize()
            windowed = rng.random() < 0.5

            def dec_rank(r):
                if not windowed:
                    m, first = groups[r].decode_sharded(comp, outs[r])
                else:
                    head = stream[:8192]
                    off, ln = groups[r].decode_window(head, comp.numel())
                    window = comp[off : off + ln].clone() if ln else None
                    # (no synchronize: the clone runs on torch's current stream, and so do the group's calls -- codec.Context follows it)
                    mine, first = groups[r].decode_begin(head, comp.numel(), window, off)
                    m = groups[r].decode_write(outs[r]) if mine else 0
                    assert m == mine
                torch.cuda.synchronize()
                return first, outs[r][:m].cpu().numpy().tobytes()

            res = run_ranks(world, dec_rank)
            if any(k == "err" for k, _ in res):
                bad += 1
                print("FAIL (decode raised)", seed, trial, world, n, src, res, flush=True)
                continue
            pieces = sorted(v for _, v in res)
            pos, ok = 0, True
            for first, piece in pieces:
                ok = ok and (first == pos or not piece)
                pos += len(piece)
            if not ok or b"".join(p for _, p in pieces) != truth:
                bad += 1
                print("FAIL (decode)", seed, trial, world, n, src, windowed, flush=True)
            elif windowed:
                win_ok += 1
            else:
                dec_ok += 1
        finally:
            for c in ctxs:
                c.close()
        if trial % 50 == 49:
            print(f"trial {trial + 1}: images {enc_ok}, decodes {dec_ok} whole + {win_ok} windowed, failures relayed {fail_ok}, bad {bad}, {time.time() - t0:.0f}s", flush=True)
    print(f"done: seed {seed}, {trials} trials, images {enc_ok}, decodes {dec_ok} whole + {win_ok} windowed, failures relayed {fail_ok}, bad {bad}, {time.time() - t0:.0f}s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

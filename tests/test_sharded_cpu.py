"""CPU, world_size 2 and 3 over gloo: the multi-GPU exchange logic of
entreepy_amd.sharded (histogram all-gather, offset plan, boundary-word merge,
concatenation, per-shard decode) with a fake compute backend standing in for the GPU.
The fake backend is the oracle -- test infrastructure; the product path has none."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from tests import corpus


class OracleBackend:
    """Implements the slice of entreepy_amd.Context that ShardedCodec calls, on CPU
    tensors, by calling the oracle."""

    def __init__(self):
        self._t = {"hist_ms": 0.0, "scan_ms": 0.0, "body_ms": 0.0}

    def timings(self, which=None):
        return dict(self._t)

    def histogram_device(self, text, hist):
        hist.copy_(torch.from_numpy(O.histogram(text.numpy()).astype(np.int64)))

    def _pack(self, cb, text, out, start_bit, header=b""):
        body, end = O.pack_body(cb.data, cb.length, text.numpy(), start_bit)
        buf = np.zeros(out.numel(), dtype=np.uint8)
        buf[: len(body)] = np.frombuffer(body, dtype=np.uint8)
        buf[: len(header)] |= np.frombuffer(header, dtype=np.uint8)
        out.copy_(torch.from_numpy(buf))
        return end

    def encode_body_device(self, cb, text, out, start_bit=0):
        return self._pack(cb, text, out, start_bit)

    def encode_head_shard_device(self, cb, text, out, header):
        return self._pack(cb, text, out, 8 * len(header), header)

    def decode_body_device(self, cb, body, n_symbols, out, start_bit=0):
        # bit-serial walk with the shard's table (the intended decoder's inner loop)
        table = {(int(cb.data[s]), int(cb.length[s])): s for s in range(256) if cb.length[s]}
        bits = np.unpackbits(body.numpy())[start_bit:]
        res, val, ln = [], 0, 0
        for b in bits:
            val, ln = (val << 1) | int(b), ln + 1
            if (val, ln) in table:
                res.append(table[(val, ln)])
                val, ln = 0, 0
                if len(res) == n_symbols:
                    break
        out[: len(res)] = torch.tensor(res, dtype=torch.uint8)
        return len(res)


    # -- cold decode of a block range (emulates et_decode_range_sync / _write) -----------
    def decode_range_sync(self, cb, stream, begin, end, in_start_bit=-1):
        table = {(int(cb.data[s]), int(cb.length[s])): s for s in range(256) if cb.length[s]}
        maxlen = int(cb.length.max())
        bits = np.unpackbits(stream.numpy())

        def step(p):  # -> (symbol or None, length); no code: skip one bit, like the kernels
            val = 0
            for ln in range(1, maxlen + 1):
                if p + ln > bits.size:
                    return None, 0
                val = (val << 1) | int(bits[p + ln - 1])
                if (val, ln) in table:
                    return table[(val, ln)], ln
            return None, 1

        if in_start_bit >= 0:
            p = begin * 8 + in_start_bit
        else:  # run in over the 128 bits in front of the range
            p = begin * 8 - 128
            while p < begin * 8:
                _, ln = step(p)
                if ln == 0:
                    break
                p += ln
        start = p - begin * 8
        syms = []
        while p < end * 8:
            sym, ln = step(p)
            if ln == 0:
                p = end * 8
                break
            if sym is not None:
                syms.append(sym)
            p += ln
        self._range_syms = syms
        return {"start_bit": start, "exit_bit": p - end * 8, "n_symbols": len(syms), "sweeps": 1}

    def decode_range_maps(self, cb, stream, begin, end, in_start_bit=-1):
        # exit of the range for every start offset (emulates et_decode_range_maps)
        self._maps_args = (cb, stream, begin, end)
        n_starts = int(cb.length.max())
        m = [0] * 32
        for p in range(32):
            if in_start_bit >= 0:
                m[p] = self.decode_range_sync(cb, stream, begin, end, in_start_bit)["exit_bit"]
            elif p < n_starts:
                m[p] = self.decode_range_sync(cb, stream, begin, end, p)["exit_bit"]
        return bytes(m), n_starts

    def decode_range_resolve(self, in_start_bit):
        cb, stream, begin, end = self._maps_args
        return self.decode_range_sync(cb, stream, begin, end, in_start_bit)

    def decode_range_write(self, max_symbols, out):
        take = self._range_syms[:max_symbols]
        out[: len(take)] = torch.tensor(take, dtype=torch.uint8)
        return len(take)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, cuts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from entreepy_amd import sharded

        data = corpus.text_like(n, 77)
        lo, hi = cuts[rank], cuts[rank + 1]
        text = torch.from_numpy(data[lo:hi].copy())
        codec = sharded.ShardedCodec(OracleBackend(), dist.group.WORLD, torch.device("cpu"))
        enc = torch.zeros(max(hi - lo, 16) + 7200 + 64, dtype=torch.uint8)
        layout = codec.encode_shard(text, enc)
        image = codec.gather_file(enc, layout)
        dec = torch.zeros(hi - lo + 64, dtype=torch.uint8)
        m = codec.decode_shard(enc, layout, dec)
        ok_dec = m == hi - lo and bool((dec[:m] == text).all())
        if rank == 0:
            q.put(("image", image))
        q.put(("dec", rank, ok_dec))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cuts", [(2, None), (3, None), (3, [0, 5, 9, 20011]), (2, [0, 0, 30000])])
def test_sharded_encode_concat_equals_single_stream(world, cuts):
    n = 30000 if cuts is None else cuts[-1]
    if cuts is None:
        cuts = [int(x) for x in np.linspace(0, n, world + 1)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, cuts, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world + 1)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    image = next(g[1] for g in got if g[0] == "image")
    assert image == O.encode(corpus.text_like(n, 77))
    assert all(g[2] for g in got if g[0] == "dec")


def test_plan_shards_offsets():
    from entreepy_amd import sharded

    data = corpus.text_like(10000, 5)
    hists = np.stack([O.histogram(data[:3000]), O.histogram(data[3000:3001]), O.histogram(data[3001:])])
    cb, header, starts = sharded.plan_shards(hists)
    od, ol, _ = O.build_dict(O.histogram(data))
    assert header == O.write_header(od, ol, len(data))
    assert starts[0] == 8 * len(header)
    _, end = O.pack_body(od, ol, data)
    assert starts[-1] - starts[0] == end
    assert sharded.owned_words(starts, 0)[0] == 0
    for r in range(2):
        assert sharded.owned_words(starts, r)[1] == sharded.owned_words(starts, r + 1)[0]


def _cold_data(n, kind):
    return corpus.text_like(n, 78) if kind == "text" else corpus.uniform(n, 79, 1, 1 + 200)


def _cold_worker(rank, world, port, n, q, kind="text"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from entreepy_amd import sharded

        data = _cold_data(n, kind)
        et = O.encode(data)
        comp = torch.from_numpy(np.frombuffer(et[4:], dtype=np.uint8).copy())
        codec = sharded.ShardedCodec(OracleBackend(), dist.group.WORLD, torch.device("cpu"))
        dec = torch.zeros(n + 64, dtype=torch.uint8)
        m, first = codec.decode_cold(comp, dec)
        q.put((rank, first, dec[:m].numpy().tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "text"), (3, "text"), (3, "flat")])
def test_cold_decode_across_ranks(world, kind):
    """sharded.decode_cold over gloo: block ranges, run-in starts, the (start, exit,
    symbols) all-gather and the repair round; pieces concatenate to the input.  "flat": a
    200-symbol uniform stream (7- and 8-bit codes only) takes the exhaustive exchange --
    per-rank start->exit maps, all-gathered and chained."""
    n = 60000 if kind == "text" else 30000  # ~35 KB / ~29 KB of body: 4-5 blocks of 8 KiB
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cold_worker, args=(r, world, port, n, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = _cold_data(n, kind).tobytes()
    pos = 0
    for rank, first, piece in got:
        assert first == pos
        pos += len(piece)
    assert b"".join(g[2] for g in got) == data

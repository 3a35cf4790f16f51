"""CPU, world_size 2 and 3 over gloo: the multi-GPU sequence of the product -- csrc/et_shard_seq.cpp, the very
file libentreepy_hip.so is built from (histogram all-gather, offset plan, boundary-word merge, concatenation,
cold decode, the failure protocol) -- with a CPU stand-in where a rank's GPU would be (tests/support/shard_cpu.cpp:
"device" memory is host memory, every compute step is the oracle's).  Test infrastructure; the product has no such
path."""
import os
import socket
import threading

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from tests import corpus


class CpuCtx:
    """What ShardedCodec asks of a context besides the group calls: the per-shard decode of the in-memory pipeline
    (et_decode_body_device), here a bit-serial walk with the shard's table -- and phase timings, here zeros."""

    def timings(self, which=None):
        return {"hist_ms": 0.0, "scan_ms": 0.0, "body_ms": 0.0}

    def decode_body_device(self, cb, body, n_symbols, out, start_bit=0):
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


def _cpu_group(rank, world, allgather):
    from entreepy_amd.codec import Group
    from tests.support import shard_cpu_lib

    return Group(CpuCtx(), rank, world, allgather=allgather, lib=shard_cpu_lib())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _codec(rank, world):
    from entreepy_amd import sharded

    return sharded.ShardedCodec(CpuCtx(), dist.group.WORLD, torch.device("cpu"), lib_group=lambda codec: _cpu_group(rank, world, codec.gather_bytes))


def _worker(rank, world, port, n, cuts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = corpus.text_like(n, 77)
        lo, hi = cuts[rank], cuts[rank + 1]
        text = torch.from_numpy(data[lo:hi].copy())
        codec = _codec(rank, world)
        enc = torch.full((max(hi - lo, 16) + 7200 + 64,), 0xFF, dtype=torch.uint8)  # dirty on purpose
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


def _spawn(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    return q, procs


def _join(procs):
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


@pytest.mark.parametrize("world,cuts", [(2, None), (3, None), (3, [0, 5, 9, 20011]), (2, [0, 0, 30000])])
def test_sharded_encode_concat_equals_single_stream(world, cuts):
    n = 30000 if cuts is None else cuts[-1]
    if cuts is None:
        cuts = [int(x) for x in np.linspace(0, n, world + 1)]
    q, procs = _spawn(_worker, world, n, cuts)
    got = [q.get(timeout=120) for _ in range(world + 1)]
    _join(procs)
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


def _cold_worker(rank, world, port, n, kind, windowed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = _cold_data(n, kind)
        et = O.encode(data)
        comp = torch.from_numpy(np.frombuffer(et[4:], dtype=np.uint8).copy())
        codec = _codec(rank, world)
        if not windowed:
            dec = torch.zeros(n + 64, dtype=torch.uint8)
            m, first = codec.decode_cold(comp, dec)
        else:
            # a rank that holds the dictionary and its own window only, and sizes its output afterwards
            g = codec.lib_group
            head = et[4 : 4 + 8192]
            off, ln = g.decode_window(head, comp.numel())
            assert off % 4 == 0 and ln <= comp.numel() // world + 8192 + 32
            window = comp[off : off + ln].clone()
            mine, first = g.decode_begin(head, comp.numel(), window, off)
            dec = torch.zeros(mine + 16, dtype=torch.uint8)
            m = g.decode_write(dec)
            assert m == mine
        q.put((rank, first, dec[:m].numpy().tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,windowed", [(2, "text", False), (3, "text", False), (3, "flat", False), (3, "text", True), (2, "flat", True)])
def test_cold_decode_across_ranks(world, kind, windowed):
    """et_decode_sharded over gloo: block ranges, run-in starts, the (start, exit,
    symbols) all-gather and the repair round; pieces concatenate to the input.  "flat": a
    200-symbol uniform stream (7- and 8-bit codes only) takes the exhaustive exchange --
    per-rank start->exit maps, all-gathered and chained.  windowed: et_decode_shard_window /
    et_decode_sharded_begin / _write on a rank that holds only its own bytes of the stream."""
    n = 60000 if kind == "text" else 30000  # ~35 KB / ~29 KB of body: 4-5 blocks of 8 KiB
    q, procs = _spawn(_cold_worker, world, n, kind, windowed)
    got = sorted(q.get(timeout=300) for _ in range(world))
    _join(procs)
    data = _cold_data(n, kind).tobytes()
    pos = 0
    for rank, first, piece in got:
        assert first == pos
        pos += len(piece)
    assert b"".join(g[2] for g in got) == data


# ---- the failure protocol: a rank that fails takes part in every exchange of the call, and all ranks return its status ----
class ThreadGather:
    """The exchange callback for ranks that are threads of one process; a rank that never arrives breaks the barrier."""

    def __init__(self, world, timeout=20):
        self.world, self.slots, self.bar, self.timeout = world, [None] * world, threading.Barrier(world), timeout
        self.calls = [0] * world

    def of(self, rank):
        def gather(mine):
            self.calls[rank] += 1
            self.slots[rank] = mine
            self.bar.wait(timeout=self.timeout)
            out = b"".join(self.slots)
            self.bar.wait(timeout=self.timeout)  # nobody overwrites a slot before all have read
            return out

        return gather


def _run_ranks(world, body):
    out, threads = [None] * world, []

    def wrap(r):
        try:
            out[r] = ("ok", body(r))
        except BaseException as e:  # noqa: BLE001 -- looked at by the main thread
            out[r] = ("err", e)

    for r in range(world):
        threads.append(threading.Thread(target=wrap, args=(r,)))
        threads[-1].start()
    for t in threads:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    return out


def _shards(world, n=40_000, seed=5):
    data = corpus.text_like(n, seed)
    cuts = [int(x) for x in np.linspace(0, n, world + 1)]
    texts = [torch.from_numpy(data[cuts[r] : cuts[r + 1]].copy()) for r in range(world)]
    encs = [torch.zeros(t.numel() + 7200 + 64, dtype=torch.uint8) for t in texts]
    return data, texts, encs


def _all_failed_with(results, status, naming_rank=None):
    from entreepy_amd.codec import EntreepyError

    for r, (kind, e) in enumerate(results):
        assert kind == "err" and isinstance(e, EntreepyError), (r, kind, e)
        assert e.status == status, (r, e)
        if naming_rank is not None:
            assert f"rank {naming_rank}" in str(e) or r == naming_rank, (r, str(e))


def test_encode_failure_on_one_rank_reaches_all():
    """world 3: rank 1's output buffer is too small for its piece / is missing -- every rank's et_encode_sharded
    returns that rank's status after the ONE exchange, nobody hangs; the group works again afterwards."""
    from entreepy_amd import _native as N

    world = 3
    data, texts, encs = _shards(world)
    x = ThreadGather(world)
    groups = [_cpu_group(r, world, x.of(r)) for r in range(world)]
    small = torch.zeros(64, dtype=torch.uint8)
    res = _run_ranks(world, lambda r: groups[r].encode_sharded(texts[r], small if r == 1 else encs[r]))
    _all_failed_with(res, N.ET_ERR_CAP, naming_rank=1)
    assert x.calls == [1, 1, 1]
    res = _run_ranks(world, lambda r: groups[r].encode_sharded(texts[r], None if r == 2 else encs[r]))
    _all_failed_with(res, N.ET_ERR_ARG, naming_rank=2)
    # all ranks empty: the reference's error.QueueEmpty, on every rank
    empty = torch.zeros(0, dtype=torch.uint8)
    res = _run_ranks(world, lambda r: groups[r].encode_sharded(empty, encs[r]))
    _all_failed_with(res, N.ET_ERR_EMPTY)
    # and a clean call on the same groups
    res = _run_ranks(world, lambda r: (groups[r].encode_sharded(texts[r], encs[r]), groups[r].merge_seams(encs[r])))
    assert all(k == "ok" for k, _ in res), res
    image = np.zeros(len(O.encode(data)) + 8, dtype=np.uint8)
    for r in range(world):
        i = groups[r].info()
        lo, hi = i["owned_word_lo"] * 4, min(i["owned_word_hi"] * 4, i["file_bytes"])
        off = (i["owned_word_lo"] - i["piece_word_lo"]) * 4
        image[lo:hi] = encs[r][off : off + hi - lo].numpy()
    assert image[: len(O.encode(data))].tobytes() == O.encode(data)


def test_merge_failure_on_one_rank_reaches_all():
    from entreepy_amd import _native as N

    world = 3
    _, texts, encs = _shards(world)
    x = ThreadGather(world)
    groups = [_cpu_group(r, world, x.of(r)) for r in range(world)]
    res = _run_ranks(world, lambda r: groups[r].encode_sharded(texts[r], encs[r]))
    assert all(k == "ok" for k, _ in res), res
    res = _run_ranks(world, lambda r: groups[r].merge_seams(None if r == 0 else encs[r]))
    _all_failed_with(res, N.ET_ERR_ARG, naming_rank=0)
    assert x.calls == [2, 2, 2]
    # a rank that calls merge without an encode of its own (its group holds no plan): the same for everybody
    y = ThreadGather(world)
    fresh = [_cpu_group(r, world, y.of(r)) for r in range(world)]
    res = _run_ranks(2, lambda r: fresh[r].encode_sharded(texts[r], encs[r]))  # rank 2 never arrives: the callback's barrier breaks
    _all_failed_with(res, N.ET_ERR_RCCL)


def test_patch_failure_behind_the_merge_exchange_strands_nobody():
    """A rank whose seam-word patch fails AFTER the merge's exchange is poisoned alone (its peers merged fine).  When all
    ranks then call merge again, nobody makes an exchange: the peers return ET_OK, the failed rank its error -- it used to
    decide by its own success, walk into the all-gather alone and hang."""
    from tests.support import shard_cpu_lib
    from entreepy_amd import _native as N
    from entreepy_amd.codec import EntreepyError

    world = 3
    _, texts, encs = _shards(world)
    x = ThreadGather(world)
    groups = [_cpu_group(r, world, x.of(r)) for r in range(world)]
    res = _run_ranks(world, lambda r: groups[r].encode_sharded(texts[r], encs[r]))
    assert all(k == "ok" for k, _ in res), res
    lib = shard_cpu_lib()
    lib.et_cpu_fail_next_patches(1)  # whichever rank patches first
    try:
        res = _run_ranks(world, lambda r: groups[r].merge_seams(encs[r]))
    finally:
        lib.et_cpu_fail_next_patches(0)
    failed = [r for r, (k, _) in enumerate(res) if k == "err"]
    assert len(failed) == 1 and res[failed[0]][1].status == N.ET_ERR_HIP, res
    calls = list(x.calls)
    res = _run_ranks(world, lambda r: groups[r].merge_seams(encs[r]))  # (hangs here without the fix: _run_ranks times out)
    assert x.calls == calls, "a repeated merge makes no exchange on any rank"
    for r, (k, e) in enumerate(res):
        if r in failed:
            assert k == "err" and isinstance(e, EntreepyError) and e.status == N.ET_ERR_HIP
        else:
            assert k == "ok"
    # a new encode clears the plan; the poisoned rank carries its status through the exchange and everybody hears of it
    res = _run_ranks(world, lambda r: groups[r].encode_sharded(texts[r], encs[r]))
    _all_failed_with(res, N.ET_ERR_HIP)


@pytest.mark.parametrize("kind", ["text", "flat"])
def test_cold_decode_failure_on_one_rank_reaches_all(kind):
    """world 3, cold decode: one rank's output buffer is too small for its share (known to all from the rows), one
    rank's stream pointer is missing, one rank's header is corrupt -- all three calls return the error, nobody hangs."""
    from entreepy_amd import _native as N

    world = 3
    data = _cold_data(60_000 if kind == "text" else 30_000, kind)
    et = O.encode(data)
    comp = torch.from_numpy(np.frombuffer(et[4:], dtype=np.uint8).copy())
    bad = comp.clone()
    bad[6] = 0  # the dictionary's first entry: a code of length 0
    outs = [torch.zeros(data.size + 64, dtype=torch.uint8) for _ in range(world)]
    x = ThreadGather(world)
    groups = [_cpu_group(r, world, x.of(r)) for r in range(world)]
    tiny = torch.zeros(16, dtype=torch.uint8)
    res = _run_ranks(world, lambda r: groups[r].decode_sharded(comp, tiny if r == 1 else outs[r]))
    _all_failed_with(res, N.ET_ERR_CAP)
    res = _run_ranks(world, lambda r: groups[r].decode_sharded(None if r == 0 else comp, outs[r]))
    _all_failed_with(res, N.ET_ERR_ARG, naming_rank=0)
    res = _run_ranks(world, lambda r: groups[r].decode_sharded(bad if r == 2 else comp, outs[r]))
    _all_failed_with(res, N.ET_ERR_FORMAT, naming_rank=2)
    # a stream too short to hold its header, an empty one: malformed, on every rank
    for short in (comp[:3], comp[:0]):
        res = _run_ranks(world, lambda r: groups[r].decode_sharded(short.clone() if short.numel() else torch.zeros(0, dtype=torch.uint8), outs[r]))
        _all_failed_with(res, N.ET_ERR_FORMAT if short.numel() else N.ET_ERR_ARG)
    # and a clean decode on the same groups
    res = _run_ranks(world, lambda r: groups[r].decode_sharded(comp, outs[r]))
    assert all(k == "ok" for k, _ in res), res
    pieces = sorted((first, outs[r][:m].numpy().tobytes()) for r, (_, (m, first)) in enumerate(res))
    assert b"".join(p for _, p in pieces) == data.tobytes()


def test_forced_collectives_world_1_callback():
    """ET_GROUP_FORCE_COLLECTIVES: a group of one goes through its transport (here the callback) instead of copying."""
    data = corpus.text_like(20_000, 3)
    text = torch.from_numpy(data.copy())
    enc = torch.zeros(data.size + 7200 + 64, dtype=torch.uint8)
    x = ThreadGather(1)
    g = _cpu_group(0, 1, x.of(0))
    g.encode_sharded(text, enc)
    g.merge_seams(enc)
    assert x.calls == [0]
    g.force_collectives(True)
    i = g.encode_sharded(text, enc)
    g.merge_seams(enc)
    assert x.calls == [2]
    assert enc[: i["file_bytes"]].numpy().tobytes() == O.encode(data)

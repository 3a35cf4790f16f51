"""GPU: every BASELINE.json config at full size, the sharded path through the C ABI
(et_shard_seq.cpp + et_shard_hip.cpp) with several ranks in one process, and the CLI's --gpus.

Config 5 (16 GiB uniform random, all 256 byte values) lies outside the reference's lossless domain:
the reference drops the most frequent symbol (Q1) and its 32-bit length field wraps (Q4), so encode
parity is what there is to check -- against the oracle on a prefix with the full stream's code table,
and through size-independent properties at full size; decode runs on the 255-symbol variant just under
4 GiB, where the format is lossless."""
import os
import subprocess
import threading

import numpy as np
import pytest

from tests import corpus
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "entreepy_amd", "entreepy")


def _rand_bytes(n, lo, hi, seed, dev):
    import torch

    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for s in range(0, n, step):
        m = min(step, n - s)
        out[s : s + m] = torch.randint(lo, hi, (m,), generator=g, device=dev, dtype=torch.int16).to(torch.uint8)
    return out


def test_config5_encode_16gib_uniform_256(ctx):
    """16 GiB, all 256 byte values: 64-bit bit offsets (~2^37), 2^18 tiles, Q1 and Q4."""
    import torch

    import entreepy_amd as E
    from oracle import oracle as O

    dev = torch.device("cuda", 0)
    n = 16 << 30
    text = _rand_bytes(n, 0, 256, 0x5EED0005, dev)
    enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device=dev)
    et_len = ctx.encode_device(text, enc)
    hist = torch.zeros(256, dtype=torch.int64, device=dev)
    ctx.histogram_device(text, hist)
    h = hist.cpu().numpy().astype(np.uint64)
    assert int(h.sum()) == n and (h > 0).all()
    cb = E.Codebook.from_histogram(h)
    od, ol, _ = O.build_dict(h)  # the oracle's code construction on the same counts
    assert np.array_equal(cb.data, od) and np.array_equal(cb.length, ol)
    assert cb.raw.n_coded == 255, "Q1: one of 256 symbols is dropped (encode.zig:57,70)"
    header = cb.header(n)
    assert header == O.write_header(od, ol, n)
    assert enc[: len(header)].cpu().numpy().tobytes() == header
    assert header[4] == 254 and header[5:9] == b"\0\0\0\0", "Q1: D byte; Q4: the 32-bit length field wraps (encode.zig:279)"
    assert et_len == len(header) + (cb.bits(h) + 7) // 8
    # the oracle on a 32 MiB prefix with the FULL stream's code table, at the image's bit phase
    pre = 32 << 20
    local = (len(header) * 8) % 32
    want, want_end = O.pack_body(cb.data, cb.length, text[:pre].cpu().numpy(), local)
    w0 = (len(header) * 8) // 32
    n_words = want_end // 32 - 1
    got = enc[w0 * 4 + 4 : (w0 + n_words) * 4].cpu().numpy().tobytes()
    assert got == want[4 : n_words * 4], "prefix of the 16 GiB image differs from the oracle's pack of the same bytes"
    # four shards at their bit offsets reproduce the single-stream image word for word
    bit = 8 * len(header)
    for r in range(4):
        view = text[r * (n // 4) : (r + 1) * (n // 4)]
        ctx.histogram_device(view, hist)
        sh = torch.zeros(view.numel() + 64, dtype=torch.uint8, device=dev)
        loc = bit % 32
        e = ctx.encode_body_device(cb, view, sh, loc)
        torch.cuda.synchronize()
        a, b = bit // 32, (bit + e - loc + 31) // 32
        assert torch.equal(sh[4 : (b - a) * 4 - 4], enc[a * 4 + 4 : b * 4 - 4]), r
        bit += e - loc
        del sh
    assert (bit + 7) // 8 == et_len


def test_config5_round_trip_uniform_255_under_4gib(ctx):
    """Bytes 1..255, 4 GiB - 1 MiB: the largest lossless stream of the format (exhaustive synchronisation)."""
    import torch

    import entreepy_amd as E

    dev = torch.device("cuda", 0)
    n = (4 << 30) - (1 << 20)
    text = _rand_bytes(n, 1, 256, 0x5EED0055, dev)
    enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device=dev)
    dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    et_len = ctx.encode_device(text, enc)
    m = ctx.decode_device(enc[4:et_len], dec)
    torch.cuda.synchronize()
    assert m == n and torch.equal(dec[:n], text)
    assert enc[5:9].cpu().numpy().tobytes() == n.to_bytes(4, "big")


def test_text_1gib_image_is_byte_exact(ctx):
    """The metric's own workload (text-1G): the whole .et image equals the CPU port's, byte for byte.
    (oracle.cpu_fast is proven equal to the restatement in test_oracle.py::test_fast_cpu_variant_*.)"""
    import torch

    import entreepy_amd as E
    from oracle import cpu_fast as F

    n = 1 << 30
    dev = torch.device("cuda", 0)
    text = corpus.text_like_torch(n, 0x5EED0004, dev)
    enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device=dev)
    et_len = ctx.encode_device(text, enc)
    want = F.encode(text.cpu().numpy(), min(len(os.sched_getaffinity(0)), 16))
    assert et_len == len(want)
    want_dev = torch.frombuffer(bytearray(want), dtype=torch.uint8).to(dev)
    assert torch.equal(enc[:et_len], want_dev)
    dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    m = ctx.decode_device(enc[4:et_len], dec)
    torch.cuda.synchronize()
    assert m == n and torch.equal(dec[:n], text)


# ---- the sharded path behind the C ABI, several ranks in one process ------------------------------
class ThreadGather:
    """The exchange callback for ranks that are threads of one process."""

    def __init__(self, world):
        self.world, self.slots, self.bar = world, [None] * world, threading.Barrier(world)

    def of(self, rank):
        def gather(mine):
            self.slots[rank] = mine
            self.bar.wait(timeout=120)
            out = b"".join(self.slots)
            self.bar.wait(timeout=120)  # nobody overwrites a slot before all have read
            return out

        return gather


def _run_ranks(world, body):
    errors, threads = [], []

    def wrap(r):
        try:
            body(r)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(f"rank {r}: {e!r}")

    for r in range(world):
        threads.append(threading.Thread(target=wrap, args=(r,)))
        threads[-1].start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    assert not errors, errors


@pytest.mark.parametrize("cuts", [
    [0, 700_001, 1_400_003, 2_000_000],          # three ordinary shards
    [0, 0, 5, 9, 9, 1_000_000],                  # empty shards (also the head), shards of a few bytes sharing one word
    [0, 300_000, 300_000, 300_001, 600_000],     # an empty shard and a one-byte shard in the middle
])
def test_c_abi_sharded_encode_concat_and_cold_decode(cuts, tmp_path):
    """et_encode_sharded -> et_shard_merge_seams -> et_shard_place / et_shard_write_fd on N ranks (threads, one
    GPU, an in-memory exchange): the image equals the oracle's, from buffers that held garbage before;
    et_decode_sharded of that image returns the text, rank by rank."""
    import torch

    import entreepy_amd as E
    from entreepy_amd.codec import Group
    from oracle import oracle as O

    dev = torch.device("cuda", 0)
    world, n = len(cuts) - 1, cuts[-1]
    data = corpus.text_like(n, 4242)
    want = O.encode(data)
    texts = [torch.from_numpy(data[cuts[r] : cuts[r + 1]].copy()).to(dev) for r in range(world)]
    encs = [torch.full((E.encode_bound(t.numel()) + 64,), 0xFF, dtype=torch.uint8, device=dev) for t in texts]  # dirty on purpose
    image = torch.full(((len(want) + 3) // 4 * 4,), 0xEE, dtype=torch.uint8, device=dev)
    path = tmp_path / "sharded.et"
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    torch.cuda.synchronize()
    x = ThreadGather(world)
    infos = [None] * world

    def encode_rank(r):
        with E.Context(0) as c:
            g = Group(c, r, world, allgather=x.of(r))
            infos[r] = g.encode_sharded(texts[r], encs[r])
            assert g.start_bits()[r] == infos[r]["start_bit"]
            g.merge_seams(encs[r])
            g.place(encs[r], image)
            g.write_fd(encs[r], fd)
            torch.cuda.synchronize()
            g.close()

    _run_ranks(world, encode_rank)
    os.close(fd)
    assert infos[0]["file_bytes"] == len(want) and infos[0]["text_len"] == n
    assert image[: len(want)].cpu().numpy().tobytes() == want, "et_shard_place: image differs from the oracle's"
    assert path.read_bytes() == want, "et_shard_write_fd: file differs from the oracle's"
    # owned ranges tile the image
    pos = 0
    for i in infos:
        assert i["owned_word_lo"] == pos
        pos = i["owned_word_hi"]
    assert pos == (len(want) * 8 + 31) // 32 or pos == (infos[-1]["end_bit"] + 31) // 32

    comp = torch.frombuffer(bytearray(want[4:]), dtype=torch.uint8).to(dev)
    outs = [torch.zeros(n + 64, dtype=torch.uint8, device=dev) for _ in range(world)]
    got = [None] * world
    torch.cuda.synchronize()
    y = ThreadGather(world)

    def decode_rank(r):
        with E.Context(0) as c:
            g = Group(c, r, world, allgather=y.of(r))
            m, first = g.decode_sharded(comp, outs[r])
            torch.cuda.synchronize()
            got[r] = (first, outs[r][:m].cpu().numpy().tobytes())
            g.close()

    _run_ranks(world, decode_rank)
    pos = 0
    for first, piece in got:
        assert first == pos
        pos += len(piece)
    assert b"".join(p for _, p in got) == data.tobytes()


def test_c_abi_sharded_all_256_symbols_and_flat_codes():
    """Uniform alphabets through the group calls: all 256 byte values (Q1: the dropped symbol emits no bits, so
    a shard of nothing else has no bits at all) and a 200-symbol flat code (the exhaustive exchange of exit
    maps in et_decode_sharded)."""
    import torch

    import entreepy_amd as E
    from entreepy_amd.codec import Group
    from oracle import oracle as O

    dev = torch.device("cuda", 0)
    base = corpus.uniform(600_000, 31, 0, 256)
    h = np.bincount(base, minlength=256)
    top = int(np.flatnonzero(h == h.max())[-1])  # the symbol the reference drops (most frequent, highest byte on ties)
    data = np.concatenate([base[:200_000], np.full(4096, top, dtype=np.uint8), base[200_000:]])
    cuts = [0, 200_000, 204_096, data.size]  # the middle shard holds nothing but the dropped symbol
    want = O.encode(data)
    world = 3
    texts = [torch.from_numpy(data[cuts[r] : cuts[r + 1]].copy()).to(dev) for r in range(world)]
    encs = [torch.full((E.encode_bound(t.numel()) + 64,), 0xFF, dtype=torch.uint8, device=dev) for t in texts]
    image = torch.zeros((len(want) + 3) // 4 * 4, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    x = ThreadGather(world)
    infos = [None] * world

    def rank(r):
        with E.Context(0) as c:
            g = Group(c, r, world, allgather=x.of(r))
            infos[r] = g.encode_sharded(texts[r], encs[r])
            g.merge_seams(encs[r])
            g.place(encs[r], image)
            torch.cuda.synchronize()
            g.close()

    _run_ranks(world, rank)
    assert infos[1]["start_bit"] == infos[1]["end_bit"], "a shard of the dropped symbol has no bits"
    assert image[: len(want)].cpu().numpy().tobytes() == want

    flat = corpus.uniform(500_000, 32, 1, 201)
    et = O.encode(flat)
    comp = torch.frombuffer(bytearray(et[4:]), dtype=torch.uint8).to(dev)
    outs = [torch.zeros(flat.size + 64, dtype=torch.uint8, device=dev) for _ in range(world)]
    got = [None] * world
    torch.cuda.synchronize()
    y = ThreadGather(world)

    def dec(r):
        with E.Context(0) as c:
            g = Group(c, r, world, allgather=y.of(r))
            m, first = g.decode_sharded(comp, outs[r])
            torch.cuda.synchronize()
            got[r] = (first, outs[r][:m].cpu().numpy().tobytes())
            g.close()

    _run_ranks(world, dec)
    assert b"".join(p for _, p in sorted(got)) == flat.tobytes()


def _statuses(world, body):
    """body(rank) on `world` threads -> per rank ("ok", value) or ("err", exception); fails if a rank hangs."""
    out, threads = [None] * world, []

    def wrap(r):
        try:
            out[r] = ("ok", body(r))
        except BaseException as e:  # noqa: BLE001 -- looked at by the caller
            out[r] = ("err", e)

    for r in range(world):
        threads.append(threading.Thread(target=wrap, args=(r,)))
        threads[-1].start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    return out


def test_c_abi_a_failing_rank_does_not_strand_the_others(tmp_path):
    """World 3 on the GPU (rank threads, in-memory exchange): a failure injected on ONE rank -- output buffer too small
    for its piece, a missing buffer, a corrupt dictionary -- comes back from ALL three calls, for et_encode_sharded,
    et_shard_merge_seams and et_decode_sharded; nobody is left waiting in an exchange, and the same groups then do a
    clean encode -> concat -> cold decode that equals the oracle's."""
    import torch

    import entreepy_amd as E
    from entreepy_amd import _native as N
    from entreepy_amd.codec import Group
    from oracle import oracle as O

    dev = torch.device("cuda", 0)
    world, n = 3, 900_000
    data = corpus.text_like(n, 515)
    want = O.encode(data)
    cuts = [0, 300_001, 600_002, n]
    texts = [torch.from_numpy(data[cuts[r] : cuts[r + 1]].copy()).to(dev) for r in range(world)]
    encs = [torch.full((E.encode_bound(t.numel()) + 64,), 0xFF, dtype=torch.uint8, device=dev) for t in texts]
    small = torch.zeros(4096, dtype=torch.uint8, device=dev)
    comp = torch.frombuffer(bytearray(want[4:]), dtype=torch.uint8).to(dev)
    bad = comp.clone()
    bad[6] = 0  # the dictionary's first entry: a code of length 0
    outs = [torch.zeros(n + 64, dtype=torch.uint8, device=dev) for _ in range(world)]
    tiny = torch.zeros(64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    x = ThreadGather(world)
    ctxs = [E.Context(0) for _ in range(world)]
    groups = [Group(ctxs[r], r, world, allgather=x.of(r)) for r in range(world)]

    def all_failed(res, status, who=None):
        for r, (kind, e) in enumerate(res):
            assert kind == "err" and isinstance(e, E.EntreepyError) and e.status == status, (r, kind, e)
            assert who is None or r == who or f"rank {who}" in str(e), (r, str(e))

    try:
        all_failed(_statuses(world, lambda r: groups[r].encode_sharded(texts[r], small if r == 1 else encs[r])), N.ET_ERR_CAP, 1)
        all_failed(_statuses(world, lambda r: groups[r].encode_sharded(texts[r], None if r == 2 else encs[r])), N.ET_ERR_ARG, 2)
        res = _statuses(world, lambda r: groups[r].encode_sharded(texts[r], encs[r]))
        assert all(k == "ok" for k, _ in res), res
        all_failed(_statuses(world, lambda r: groups[r].merge_seams(None if r == 0 else encs[r])), N.ET_ERR_ARG, 0)
        res = _statuses(world, lambda r: groups[r].merge_seams(encs[r]))
        assert all(k == "ok" for k, _ in res), res
        image = torch.zeros((len(want) + 3) // 4 * 4, dtype=torch.uint8, device=dev)
        for r in range(world):
            groups[r].place(encs[r], image)
        torch.cuda.synchronize()
        assert image[: len(want)].cpu().numpy().tobytes() == want
        all_failed(_statuses(world, lambda r: groups[r].decode_sharded(comp, tiny if r == 1 else outs[r])), N.ET_ERR_CAP, 1)
        all_failed(_statuses(world, lambda r: groups[r].decode_sharded(None if r == 0 else comp, outs[r])), N.ET_ERR_ARG, 0)
        all_failed(_statuses(world, lambda r: groups[r].decode_sharded(bad if r == 2 else comp, outs[r])), N.ET_ERR_FORMAT, 2)
        res = _statuses(world, lambda r: groups[r].decode_sharded(comp, outs[r]))
        assert all(k == "ok" for k, _ in res), res
        torch.cuda.synchronize()
        pieces = sorted((first, outs[r][:m].cpu().numpy().tobytes()) for r, (_, (m, first)) in enumerate(res))
        assert b"".join(p for _, p in pieces) == data.tobytes()
    finally:
        for c in ctxs:
            c.close()  # (closes its group first)


def test_forced_collectives_run_every_rccl_call_on_one_gpu():
    """ET_GROUP_FORCE_COLLECTIVES on an RCCL group of ONE: the histogram rows go through ncclAllGather from device memory
    and the kernel that hands them to the polling host, the seam and cold-decode rows through the generic RCCL exchange,
    et_shard_gather sends and receives the piece to itself -- the calls an N-GPU run makes, on the one GPU of the box.
    The image equals the oracle's; the cold decode returns the text."""
    import torch

    import entreepy_amd as E
    from entreepy_amd import _native as N
    from entreepy_amd.codec import Group
    from oracle import oracle as O

    dev = torch.device("cuda", 0)
    assert b"rccl" in N.lib().et_rccl_library()
    with E.Context(0) as c:
        c.use_torch_stream()
        g = Group(c, 0, 1, rccl_id=Group.rccl_unique_id())
        g.force_collectives(True)
        g.set_timeout_ms(30_000)
        for data in (corpus.text_like(3_000_001, 41), corpus.uniform(700_000, 72, 1, 201), np.frombuffer(b"hello, world", dtype=np.uint8)):
            want = O.encode(data)
            text = torch.from_numpy(data.copy()).to(dev)
            enc = torch.full((E.encode_bound(data.size) + 64,), 0xFF, dtype=torch.uint8, device=dev)
            for _ in range(3):  # (the same rows again: the status / cap words stay on the device)
                i = g.encode_sharded(text, enc)
            assert i["file_bytes"] == len(want) and i["exchange_ms"] > 0
            g.merge_seams(enc)
            image = torch.full(((len(want) + 3) // 4 * 4,), 0xEE, dtype=torch.uint8, device=dev)
            g.gather(enc, image, 0)
            torch.cuda.synchronize()
            assert image[: len(want)].cpu().numpy().tobytes() == want
            comp = torch.frombuffer(bytearray(want[4:]), dtype=torch.uint8).to(dev)
            out = torch.zeros(data.size + 64, dtype=torch.uint8, device=dev)
            m, first = g.decode_sharded(comp, out)
            torch.cuda.synchronize()
            assert first == 0 and out[:m].cpu().numpy().tobytes() == data.tobytes()
        # a failure still reaches the caller through the forced exchange
        with pytest.raises(E.EntreepyError) as e:
            g.encode_sharded(text, torch.zeros(4, dtype=torch.uint8, device=dev))
        assert e.value.status == N.ET_ERR_CAP
        g.close()


def test_cli_gpus_shards_one_file(tmp_path, res_files):
    """`entreepy --gpus N c/d`: one file over N ranks (threads of the CLI; ranks share the box's one GPU) --
    the same bytes as the single-GPU CLI and the oracle, both directions."""
    from oracle import oracle as O

    text = corpus.tiled_midsummer(3_000_000).tobytes()
    src = tmp_path / "in.txt"
    src.write_bytes(text)
    want = O.encode(np.frombuffer(text, dtype=np.uint8))
    for gpus in (2, 5):
        et = tmp_path / f"out{gpus}.et"
        r = subprocess.run([EXE, "--gpus", str(gpus), "c", str(src), "-o", str(et)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert et.read_bytes() == want
        assert r.stderr.strip() == f"{O.format_file_size(len(text))} => {O.format_file_size(len(want))}"
        back = tmp_path / f"back{gpus}.txt"
        r = subprocess.run([EXE, "--gpus", str(gpus), "-d", "d", str(et), "-o", str(back)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert back.read_bytes() == text
        # every rank read its own 8 KiB-block range with 16 bytes on either side, not the file (the dictionary is read once, by the process)
        reads = [int(ln.split()[3]) for ln in r.stdout.splitlines() if ln.startswith("rank ")]
        assert len(reads) == gpus and all(0 < b <= (len(want) - 4) // gpus + 8192 + 32 for b in reads), reads
        assert sum(reads) <= len(want) - 4 + 32 * gpus
    # a tiny file: more ranks than bytes worth a word
    small = tmp_path / "small.txt"
    small.write_bytes(res_files["test.txt"])
    et = tmp_path / "small.et"
    r = subprocess.run([EXE, "--gpus", "4", "c", str(small), "-o", str(et)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert et.read_bytes() == O.encode(res_files["test.txt"])


def test_cli_gpus_a_ranks_own_failure_decides_message_and_exit_status(tmp_path, res_files):
    """`entreepy --gpus 3` with one rank's preparation failing (ET_CLI_TEST_FAIL_RANK: as an unreadable input range would):
    that rank still makes its group calls with nothing to contribute, its peers hear ET_ERR_ARG through the exchange rows --
    and what the process reports is the failing rank's OWN cause (an I/O error reading the input), not the echo; nobody hangs,
    for compress and for decompress."""
    src = tmp_path / "t.txt"
    src.write_bytes(res_files["a_midsummer_nights_dream.txt"] * 3)
    et = tmp_path / "t.et"
    r = subprocess.run([EXE, "--gpus", "3", "c", str(src), "-o", str(et)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for mode, args in (("c", [str(src), "-o", str(tmp_path / "x.et")]), ("d", [str(et), "-o", str(tmp_path / "x.txt")])):
        for rank in (0, 1, 2):
            r = subprocess.run([EXE, "--gpus", "3", mode] + args, capture_output=True, text=True, timeout=300, env=dict(os.environ, ET_CLI_TEST_FAIL_RANK=str(rank)))
            assert r.returncode == 1, (mode, rank, r.stderr)
            assert "reading the input" in r.stderr and "file read/write error" in r.stderr, (mode, rank, r.stderr)


def test_cli_debug_prefix_collision_check(tmp_path):
    """-d also runs the reference's prefix self-check (encode.zig:221-247; the loop is unit-tested on a
    crafted table in test_host_logic.py).  A Huffman table never trips it -- not even one the u32 truncation
    has damaged: a Fibonacci-like histogram with code lengths up to 35 -- so -d prints no such line."""
    from oracle import oracle as O

    fib = [1, 1]
    while len(fib) < 36:
        fib.append(fib[-1] + fib[-2])
    data = np.concatenate([np.full(c, 65 + i, dtype=np.uint8) for i, c in enumerate(fib)])
    src = tmp_path / "fib.bin"
    src.write_bytes(data.tobytes())
    _, ol, _ = O.build_dict(O.histogram(data))
    assert int(ol.max()) == 35
    r = subprocess.run([EXE, "-dt", "c", str(src)], capture_output=True, timeout=300)
    assert r.returncode == 0 and b"bits in output" in r.stdout and b"Found colliding" not in r.stdout
    t = tmp_path / "t.txt"
    t.write_bytes(corpus.tiled_midsummer(200_000).tobytes())
    r = subprocess.run([EXE, "-dt", "c", str(t)], capture_output=True, timeout=300)
    assert r.returncode == 0 and b"Found colliding" not in r.stdout

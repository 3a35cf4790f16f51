"""GPU: the row walk (csrc/et_rowsync.hip) -- one-pass synchronisation of complete codes of 7 and 8 bits, the streams
BASELINE.json calls its worst case (uniform bytes) -- against the oracle, through the C ABI.

What a decode runs is asserted (timings()["row_sync"]), and so is the fallback for the same streams (ET_NO_ROW_SYNC=1 in a
child process: the exit maps of et_kernels_fallback.hip), so both stay pinned."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as O

    return O


def flat(k, n, seed, lo=None):
    """n bytes over k distinct values with (almost) equal counts, shuffled: lengths 7 and 8 for 129 <= k <= 255."""
    rng = np.random.default_rng(seed)
    lo = (256 - k) // 2 if lo is None else lo
    vals = (np.arange(k) + lo).astype(np.uint8)
    data = np.tile(vals, n // k + 1)[:n]
    rng.shuffle(data)
    return data


def _timed_decode(ctx, et):
    ctx.enable_timing(True)
    try:
        back = ctx.decode(et[4:])
        t = ctx.timings("decode")
    finally:
        ctx.enable_timing(False)
    return back, t


@pytest.mark.parametrize("k", [129, 130, 135, 139, 215, 240, 250, 254, 255])
def test_flat_alphabets_decode_by_rows(ctx, k):
    """T = 256 - k seven-bit codes, from 127 down to 1: paths that change column at almost every row and paths that never do.
    (Between ~140 and ~210 symbols such a code settles quickly enough for the tree walk, which is then the faster one:
    tests/test_gpu_fixedsync.py::test_codes_of_two_lengths_...; the row walk on those alphabets: the hand-made t = 64 below behind
    ET_NO_ROW_SYNC's sibling switch, test_row_walk_on_codes_the_tree_walk_would_take.)"""
    import entreepy_amd as E

    O = _oracle()
    data = flat(k, 300_007, 40 + k)
    want = O.encode(data.tobytes())
    cb, n, off = E.parse_header(want[4:])
    assert cb.raw.max_length == 8 and cb.raw.min_length == 7 and cb.raw.n_coded == k
    assert ctx.encode(data.tobytes()) == want
    back, t = _timed_decode(ctx, want)
    assert t["row_sync"] and t["exhaustive_sync"] and t["chained_write"]
    assert back == data.tobytes()


def test_sizes_around_subsequences_blocks_and_chunks(ctx):
    """Stream ends everywhere around a 256-bit subsequence, an 8 KiB block and a ticket's chunk of blocks (the lanes the
    stream ends in walk one codeword at a time; chunks and blocks may be partly or wholly empty).  The code is fixed
    (255 symbols of equal weight), the body packed by the oracle, so any number of symbols will do."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    h = np.zeros(256, dtype=np.uint64)
    h[1:] = 100
    cb = E.Codebook.from_histogram(h)
    assert cb.raw.max_length == 8 and cb.raw.min_length == 7 and cb.raw.n_coded == 255
    chunk_text = 4 * 8192  # ~symbols per chunk (codes are ~8 bits)
    sizes = [1, 2, 3, 4, 5, 31, 32, 33, 34, 63, 64, 65, 255, 256, 257, 8190, 8192, 8200, 8224, 8225, 8226, 8192 * 2 + 5, chunk_text - 40, chunk_text - 1, chunk_text,
             chunk_text + 1, chunk_text + 129, chunk_text + 300, 2 * chunk_text + 17, 5 * chunk_text - 3, 1_000_003]
    sizes += list(range(8192 - 12, 8192 + 48))  # every end position around the first block's end (7.97 bits per symbol)
    base = flat(255, max(sizes) + 1000, 7, lo=1)
    out = torch.empty(max(sizes) + 64, dtype=torch.uint8, device="cuda")
    for n in sizes:
        data = base[:n]
        body, end_bit = O.pack_body(cb.data, cb.length, data, 0)
        buf = torch.frombuffer(bytearray(body) + bytearray(64), dtype=torch.uint8).cuda()
        ctx.enable_timing(True)
        try:
            m = ctx.decode_body_device(cb, buf[: (end_bit + 7) // 8], n, out, 0)
            t = ctx.timings("decode")
        finally:
            ctx.enable_timing(False)
        assert t["row_sync"], n
        assert m == n and out[:m].cpu().numpy().tobytes() == data.tobytes(), n


def test_every_truncation_of_a_small_stream(ctx):
    """Truncated streams: the codeword the stream's end cuts is nobody's, wherever in a row, a subsequence or a block the
    end falls (the oracle's intended decoder decides)."""
    import entreepy_amd as E

    O = _oracle()
    data = flat(200, 9_000, 3)
    et = O.encode(data.tobytes())[4:]
    _, _, off = E.parse_header(et)
    checked = 0
    for cut in list(range(off + 1, off + 80)) + list(range(len(et) - 300, len(et))) + list(range(off + 8100, off + 8300, 3)):
        part = et[:cut]
        assert ctx.decode(part) == O.decode(part), cut
        checked += 1
    assert checked > 400


def test_body_alignments_and_start_bits(ctx):
    """The body from any byte offset (its 4-byte aligned base lies up to 3 bytes before it) and any start bit: the stream's
    first codeword begins at bit first_bit < 32 of the first subsequence."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    data = flat(255, 70_001, 11, lo=1)
    h = np.bincount(data, minlength=256).astype(np.uint64)
    cb = E.Codebook.from_histogram(h)
    assert cb.raw.max_length == 8 and cb.raw.min_length == 7
    for start_bit in range(8):
        body, end_bit = O.pack_body(cb.data, cb.length, data, start_bit)
        for shift in range(4):
            buf = torch.zeros(len(body) + 64, dtype=torch.uint8, device="cuda")
            buf[16 + shift : 16 + shift + len(body)] = torch.frombuffer(bytearray(body), dtype=torch.uint8).cuda()
            out = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
            ctx.enable_timing(True)
            try:
                m = ctx.decode_body_device(cb, buf[16 + shift : 16 + shift + (end_bit + 7) // 8], data.size, out, start_bit)
                t = ctx.timings("decode")
            finally:
                ctx.enable_timing(False)
            assert t["row_sync"]
            assert m == data.size and out[:m].cpu().numpy().tobytes() == data.tobytes(), (start_bit, shift)


def test_declared_length_shorter_than_the_body(ctx):
    """The declared symbol count is what a decode returns at most: a count that ends inside a wavefront's 64 subsequences, on a
    16-byte boundary of the output, one symbol, and more than the body holds (the pad bits then decode as whatever they say)."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    h = np.zeros(256, dtype=np.uint64)
    h[1:] = 100
    cb = E.Codebook.from_histogram(h)
    data = flat(255, 200_000, 21, lo=1)
    body, end_bit = O.pack_body(cb.data, cb.length, data, 0)
    buf = torch.frombuffer(bytearray(body) + bytearray(64), dtype=torch.uint8).cuda()
    out = torch.full((data.size + 256,), 0xEE, dtype=torch.uint8, device="cuda")
    for n_decl in (1, 15, 16, 17, 2047, 2048, 8191, 8192 * 3 + 5, 100_000, data.size - 1, data.size):
        out.fill_(0xEE)
        m = ctx.decode_body_device(cb, buf[: (end_bit + 7) // 8], n_decl, out, 0)
        got = out.cpu().numpy()
        assert m == n_decl and got[:m].tobytes() == data[:m].tobytes(), n_decl
        assert (got[m + 16 :] == 0xEE).all(), f"bytes written far behind the {n_decl} declared symbols"
    # more declared than there are codewords: every codeword of the body, pad bits included, and no more
    m = ctx.decode_body_device(cb, buf[: (end_bit + 7) // 8], data.size + 100, out, 0)
    assert data.size <= m <= data.size + 1 and out[: data.size].cpu().numpy().tobytes() == data.tobytes()


def test_128_equal_symbols_are_a_row_code_of_7_bit_codewords_only(ctx):
    """The reference's builder on 128 symbols of equal weight: a complete tree of depth 7 -- t = 128, no 8-bit codeword at all;
    every codeword moves the walk a column to the left and every eighth one wraps into the same row."""
    import ctypes

    import entreepy_amd as E
    from entreepy_amd import _native as N

    O = _oracle()
    rng = np.random.default_rng(128)
    data = np.repeat(rng.permutation(256)[:128].astype(np.uint8), 2500)
    rng.shuffle(data)
    want = O.encode(data.tobytes())
    cb, n, off = E.parse_header(want[4:])
    assert cb.raw.max_length == 7 and cb.raw.min_length == 7 and cb.raw.n_coded == 128
    got_t = ctypes.c_uint32(999)
    assert N.lib().et_row_code(ctypes.byref(cb.raw), ctypes.byref(got_t)) == N.ET_OK and got_t.value == 128
    assert ctx.encode(data.tobytes()) == want
    back, t = _timed_decode(ctx, want)
    assert t["fixed_sync"] and t["exhaustive_sync"]  # (a fixed-length code first of all: tests/test_gpu_fixedsync.py; by rows in the child below)
    assert back == data.tobytes()
    code = (
        "import numpy as np, entreepy_amd as E\n"
        "from oracle import oracle as O\n"
        "c = E.Context(0); c.enable_timing(True)\n"
        "rng = np.random.default_rng(128)\n"
        "d = np.repeat(rng.permutation(256)[:128].astype(np.uint8), 2500); rng.shuffle(d)\n"
        "for cut in (0, 1, 7, 33):\n"
        "    et = O.encode(d.tobytes())\n"
        "    et = et[: len(et) - cut]\n"
        "    assert c.decode(et[4:]) == O.decode(et[4:])\n"
        "    t = c.timings('decode')\n"
        "    assert t['row_sync'] and not t['fixed_sync'], t\n"
        "print('ok')\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, ET_NO_FIXED_SYNC="1"), timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("t", [0, 1, 5, 64, 127, 128])
def test_hand_made_row_dictionaries(ctx, t):
    """Row codes that no encoder made: the 7-bit codewords 0 .. t-1 and the 8-bit codewords 2t .. 255 handed to symbols in random
    order (t = 0: all 256 byte values with 8 bits each, which the reference's own encoder cannot produce -- it drops one of 256, Q1;
    t = 128: 128 symbols of 7 bits, every row a step to the left).
    The body is packed by the oracle with that table; the decode must return the text (the symbol table of the write pass is filled
    from the dictionary, not from an order the encoder would have used)."""
    import ctypes

    import torch

    import entreepy_amd as E
    from entreepy_amd import _native as N

    O = _oracle()
    rng = np.random.default_rng(900 + t)
    n_sym = 256 - t
    syms = rng.permutation(256)[:n_sym]
    data_t = np.zeros(256, dtype=np.uint32)
    len_t = np.zeros(256, dtype=np.uint8)
    codes = [(v, 7) for v in range(t)] + [(v, 8) for v in range(2 * t, 256)]
    assert len(codes) == n_sym
    for s, (v, ln) in zip(syms, [codes[i] for i in rng.permutation(n_sym)]):
        data_t[s], len_t[s] = v, ln
    cb = E.Codebook.from_tables(data_t, len_t)
    got_t = ctypes.c_uint32(999)
    assert N.lib().et_row_code(ctypes.byref(cb.raw), ctypes.byref(got_t)) == N.ET_OK and got_t.value == t
    text = syms[rng.integers(0, n_sym, size=150_001)].astype(np.uint8)
    for start_bit in (0, 3):
        body, end_bit = O.pack_body(cb.data, cb.length, text, start_bit)
        buf = torch.frombuffer(bytearray(body) + bytearray(64), dtype=torch.uint8).cuda()
        out = torch.empty(text.size + 64, dtype=torch.uint8, device="cuda")
        ctx.enable_timing(True)
        try:
            m = ctx.decode_body_device(cb, buf[: (end_bit + 7) // 8], text.size, out, start_bit)
            tm = ctx.timings("decode")
        finally:
            ctx.enable_timing(False)
        # (t = 0 and t = 128 are fixed-length codes, which go by arithmetic: tests/test_gpu_fixedsync.py)
        # (... and t = 64 -- half of the 7-bit patterns are codewords -- settles within the tree walk's reach, which takes it)
        assert tm["fixed_sync" if t in (0, 128) else ("tree_walk_sync" if t == 64 else "row_sync")] and m == text.size and out[:m].cpu().numpy().tobytes() == text.tobytes(), (t, start_bit)


def test_fuzzed_bodies_match_the_oracle(ctx):
    """Bit flips, junk and all-ones / all-zeros runs in the body: any bit pattern is a codeword of a complete code, so the
    decode is whatever the bits say -- and equals the oracle's."""
    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(99)
    for trial in range(10):
        k = int(rng.choice([129, 180, 254, 255]))
        data = flat(k, int(rng.integers(30_000, 300_000)), 100 + trial)
        good = bytearray(O.encode(data.tobytes())[4:])
        _, _, off = E.parse_header(bytes(good))
        kind = trial % 4
        if kind == 0:
            for pos in rng.integers(off, len(good), size=60):
                good[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            a = int(rng.integers(off, len(good) - 5000))
            good[a : a + 4096] = rng.integers(0, 256, size=4096, dtype=np.uint8).tobytes()
        elif kind == 2:
            del good[int(rng.integers(off + 1, len(good))) :]
        else:
            good[-9000:-4500] = b"\xff" * 4500
            good[-4500:] = b"\x00" * 4500
        assert ctx.decode(bytes(good)) == O.decode(bytes(good)), (trial, kind, k)


def test_large_stream_many_chunks_in_flight(ctx):
    """256 MiB of 255 uniform byte values: ~8000 chunks, thousands in flight at once, look-backs over whatever has been
    published -- an exact round trip, twice (the second run finds the scratch words of the first)."""
    import torch

    import entreepy_amd as E

    n = 256 << 20
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    text = torch.randint(1, 256, (n,), generator=g, device="cuda", dtype=torch.int16).to(torch.uint8)
    enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device="cuda")
    dec = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    et_len = ctx.encode_device(text, enc)
    for _ in range(2):
        dec.zero_()
        ctx.enable_timing(True)
        try:
            m = ctx.decode_device(enc[4:et_len], dec)
            t = ctx.timings("decode")
        finally:
            ctx.enable_timing(False)
        assert t["row_sync"]
        assert m == n and torch.equal(dec[:n], text)


@pytest.mark.parametrize("k,ranks", [(255, 2), (255, 7), (200, 3), (130, 5), (128, 3)])
def test_ranges_of_a_stream_split_over_ranks(k, ranks):
    """et_decode_range_maps / _resolve / _write on a row code's stream cut into block ranges, each on its own et_ctx (what
    et_decode_sharded does on N GPUs): a range's MAP -- entry column -> the column the next range is entered in -- comes from one
    pass in which every chunk publishes its map and the last one composes them; the maps are chained by hand here as the group's
    exchange chains them, every range is resolved from its true start and written by rows; the pieces are the oracle's decode."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    data = flat(k, 700_001 + 1000 * ranks, 500 + k)
    et = O.encode(data.tobytes())
    comp = torch.from_numpy(np.frombuffer(et[4:], dtype=np.uint8).copy()).cuda()
    cb, n_symbols, body_off = E.parse_header(et[4:])
    ptr = comp.data_ptr() + body_off
    base_off, first_bit = body_off - (ptr & 3), (ptr & 3) * 8
    stream = comp[base_off:]
    n_blocks = (stream.numel() + 8191) // 8192
    ctxs, maps = [], []
    try:
        for r in range(ranks):
            lo, hi = r * n_blocks // ranks, (r + 1) * n_blocks // ranks
            c = E.Context(0)
            begin, end = lo * 8192, min(hi * 8192, stream.numel())
            m, n_starts = c.decode_range_maps(cb, stream, begin, end, first_bit if lo == 0 else -1)
            assert n_starts == 8
            if lo == 0:
                assert len(set(m[:8])) == 1, "the stream's first range begins at a known bit: a constant map"
            ctxs.append(c)
            maps.append(m)
        out, first, s_in = [], 0, first_bit
        for c, m in zip(ctxs, maps):
            inf = c.decode_range_resolve(s_in)
            assert inf["row_walk"] and inf["start_bit"] == s_in and inf["exit_bit"] == m[s_in], (inf, m[:8], s_in)
            s_in = m[s_in]
            take = max(0, min(inf["n_symbols"], n_symbols - first))
            buf = torch.empty(inf["n_symbols"] + 64, dtype=torch.uint8, device="cuda")
            got = c.decode_range_write(take, buf)
            torch.cuda.synchronize()
            out.append(buf[:got].cpu().numpy())
            first += inf["n_symbols"]
        assert np.concatenate(out).tobytes() == data.tobytes()
    finally:
        for c in ctxs:
            c.close()


def test_row_walk_on_codes_the_tree_walk_would_take():
    """ET_NO_QUICK_SYNC=1 (a child process): 160 and 200 symbols -- t = 96 and 56 seven-bit codewords, paths that change column
    every few rows -- by rows, as before the decode learnt to try the tree walk on them."""
    code = (
        "import numpy as np, entreepy_amd as E\n"
        "from tests.test_gpu_rowsync import flat\n"
        "from oracle import oracle as O\n"
        "c = E.Context(0); c.enable_timing(True)\n"
        "for k in (160, 200):\n"
        "    d = flat(k, 400_003, k)\n"
        "    et = O.encode(d.tobytes())\n"
        "    assert c.decode(et[4:]) == d.tobytes()\n"
        "    t = c.timings('decode')\n"
        "    assert t['row_sync'] and not t['tree_walk_sync'], t\n"
        "    for cut in (1, 5, 8191):\n"
        "        assert c.decode(et[4:-cut]) == O.decode(et[4:-cut])\n"
        "print('ok')\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, ET_NO_QUICK_SYNC="1"), timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("switch", ["ET_NO_ROW_SYNC", "ET_NO_ROW_WRITE"])
def test_fallback_exit_maps_still_decode_these_streams(switch):
    """ET_NO_ROW_SYNC=1 (a child process: the switch is read once): the same streams through k_dec_maps_reg / k_dec_resolve_reg
    and the chained-table write; ET_NO_ROW_WRITE=1: the row walk's synchronisation with the chained-table write behind it."""
    code = (
        "import numpy as np, entreepy_amd as E\n"
        "from tests.test_gpu_rowsync import flat\n"
        "from oracle import oracle as O\n"
        "c = E.Context(0); c.enable_timing(True)\n"
        "for k in (130, 255):\n"
        "    d = flat(k, 400_003, k)\n"
        "    et = O.encode(d.tobytes())\n"
        "    assert c.decode(et[4:]) == d.tobytes()\n"
        "    t = c.timings('decode')\n"
        "    assert t['exhaustive_sync'] and t['row_sync'] == (SWITCH == 'ET_NO_ROW_WRITE'), t\n"
        "print('ok')\n"
    ).replace("SWITCH", repr(switch))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, **{switch: "1"}), timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]

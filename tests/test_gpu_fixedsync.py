"""GPU: fixed-length codes (2^L codewords of L bits -- 4, 8, 16, 32, 64 symbols of about equal weight) are synchronised by
arithmetic (k_fixed_sync, csrc/et_rowsync.hip): against the oracle, through the C ABI; what a decode runs is asserted, and the
exit maps that decoded these streams before stay pinned behind ET_NO_FIXED_SYNC=1."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT
from tests.test_gpu_rowsync import _oracle, _timed_decode, flat

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [2, 4, 8, 16, 32, 64, 128])
def test_power_of_two_alphabets_decode_by_arithmetic(ctx, k):
    """The reference's builder on 2^L symbols of equal weight gives 2^L codewords of L bits; .et images equal the oracle's,
    the decode takes k_fixed_sync and k_fixed_write."""
    import entreepy_amd as E

    O = _oracle()
    L = k.bit_length() - 1
    data = flat(k, 300_007, 60 + k)
    want = O.encode(data.tobytes())
    cb, n, off = E.parse_header(want[4:])
    assert cb.raw.max_length == L and cb.raw.min_length == L and cb.raw.n_coded == k
    assert ctx.encode(data.tobytes()) == want
    back, t = _timed_decode(ctx, want)
    assert t["fixed_sync"] and t["exhaustive_sync"] and not t["row_sync"]
    assert back == data.tobytes()


@pytest.mark.parametrize("k,quick", [(3, True), (6, True), (10, True), (17, True), (26, True), (36, True), (50, True), (100, True), (31, False), (62, False), (120, False), (65, False), (160, True), (200, True), (136, False), (240, False)])
def test_codes_of_two_lengths_try_the_tree_walk_when_they_settle_quickly(ctx, k, quick):
    """k symbols of equal weight, k not a power of two: codewords of L and L + 1 bits.  Most such codes re-synchronise within the tree
    walk's reach (et::quick_to_synchronise estimates it from the share of short codewords) and decode like text; those with a lone short
    or long codeword among many (k = 2^L + 1, 2^(L+1) - 1) do not and go to the exit maps at once.  Either way: the oracle's bytes."""
    O = _oracle()
    data = flat(k, 600_011, 300 + k)
    et = O.encode(data.tobytes())
    assert ctx.encode(data.tobytes()) == et
    back, t = _timed_decode(ctx, et)
    assert back == data.tobytes()
    assert t["tree_walk_sync"] == quick and t["exhaustive_sync"] == (not quick), t
    # ... and et_decode_path, the diagnostic, says what ran
    import ctypes

    import entreepy_amd as E
    from entreepy_amd import _native as N

    cb, _, _ = E.parse_header(et[4:])
    p = ctypes.c_uint32(99)
    assert N.lib().et_decode_path(ctypes.byref(cb.raw), ctypes.byref(p)) == N.ET_OK
    assert p.value == (N.ET_PATH_TREE_WALK if quick else (N.ET_PATH_ROWS if t["row_sync"] else N.ET_PATH_EXIT_MAPS))
    for cut in (1, 3, 8190, 8200):
        assert ctx.decode(et[4:-cut]) == O.decode(et[4:-cut]), cut


def test_almost_flat_alphabets_take_their_own_paths(ctx):
    """10 symbols (3 and 4 bits) or 3 (1 and 2 bits) are not a fixed-length code: the exit maps / the tree walk."""
    O = _oracle()
    for k, want_fixed in ((3, False), (10, False), (16, True)):
        data = flat(k, 100_003, k)
        et = O.encode(data.tobytes())
        back, t = _timed_decode(ctx, et)
        assert back == data.tobytes() and t["fixed_sync"] == want_fixed, k


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 6, 7])
def test_sizes_start_bits_and_alignments(ctx, L):
    """Bodies packed by the oracle from any start bit and byte alignment, ending everywhere around a subsequence and a block
    (L = 3, 5, 6: a codeword straddles the subsequences' seams at a different offset in every one of them)."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    k = 1 << L
    h = np.zeros(256, dtype=np.uint64)
    h[10 : 10 + k] = 50
    cb = E.Codebook.from_histogram(h)
    assert cb.raw.max_length == L and cb.raw.min_length == L
    per_sub, per_block = 256 // L, 65536 // L
    sizes = [1, 2, per_sub - 1, per_sub, per_sub + 1, per_sub + 2, 2 * per_sub + 1, per_block - 2, per_block - 1, per_block, per_block + 1, per_block + 2, 3 * per_block + 7, 500_001]
    base = flat(k, max(sizes) + 100, 77 + L, lo=10)
    out = torch.empty(max(sizes) + 256, dtype=torch.uint8, device="cuda")
    for n in sizes:
        data = base[:n]
        for start_bit, shift in ((0, 0), (1, 1), (5, 2), (7, 3), (3, 0)):
            body, end_bit = O.pack_body(cb.data, cb.length, data, start_bit)
            buf = torch.zeros(len(body) + 64, dtype=torch.uint8, device="cuda")
            buf[16 + shift : 16 + shift + len(body)] = torch.frombuffer(bytearray(body), dtype=torch.uint8).cuda()
            ctx.enable_timing(True)
            try:
                m = ctx.decode_body_device(cb, buf[16 + shift : 16 + shift + (end_bit + 7) // 8], n, out, start_bit)
                t = ctx.timings("decode")
            finally:
                ctx.enable_timing(False)
            assert t["fixed_sync"], (n, start_bit)
            assert m == n and out[:m].cpu().numpy().tobytes() == data.tobytes(), (n, start_bit, shift)


def test_truncations_and_declared_lengths(ctx):
    """Truncated images (a codeword the stream's end cuts is nobody's; whole codewords in the pad bits are decoded, as the oracle's
    intended decoder does) and declared counts shorter than the body."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    for k in (4, 64):
        data = flat(k, 40_000, 5 + k)
        et = O.encode(data.tobytes())[4:]
        cb, _, off = E.parse_header(et)
        for cut in list(range(off + 1, off + 40)) + list(range(len(et) - 120, len(et) + 1)) + list(range(off + 8180, off + 8210)):
            part = et[:cut]
            assert ctx.decode(part) == O.decode(part), (k, cut)
        body, end_bit = O.pack_body(cb.data, cb.length, data, 0)
        buf = torch.frombuffer(bytearray(body) + bytearray(64), dtype=torch.uint8).cuda()
        out = torch.full((data.size + 256,), 0xEE, dtype=torch.uint8, device="cuda")
        for n_decl in (1, 15, 16, 17, 2047, 8192 * 3 + 5, data.size - 1, data.size):
            out.fill_(0xEE)
            m = ctx.decode_body_device(cb, buf[: (end_bit + 7) // 8], n_decl, out, 0)
            got = out.cpu().numpy()
            assert m == n_decl and got[:m].tobytes() == data[:m].tobytes(), (k, n_decl)
            assert (got[m + 16 :] == 0xEE).all(), (k, n_decl)
        # more declared than the body holds: every whole codeword, the pad bits' included, and no more
        m = ctx.decode_body_device(cb, buf[: (end_bit + 7) // 8], data.size + 100, out, 0)
        assert data.size <= m <= data.size + 7 // cb.raw.max_length and out[: data.size].cpu().numpy().tobytes() == data.tobytes()


def test_hand_made_fixed_length_dictionary(ctx):
    """Sixteen 4-bit codewords handed to symbols in an order no encoder would choose; with one of the sixteen missing the code is
    not complete and the decode must NOT take the arithmetic (a bit pattern without a symbol is the fallback's business)."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(16)
    syms = rng.permutation(256)[:16]
    for missing in (False, True):
        data_t = np.zeros(256, dtype=np.uint32)
        len_t = np.zeros(256, dtype=np.uint8)
        use = syms[:15] if missing else syms
        for s, v in zip(use, rng.permutation(16)):
            data_t[s], len_t[s] = v, 4
        cb = E.Codebook.from_tables(data_t, len_t)
        text = use[rng.integers(0, len(use), size=90_001)].astype(np.uint8)
        body, end_bit = O.pack_body(cb.data, cb.length, text, 3)
        buf = torch.frombuffer(bytearray(body) + bytearray(64), dtype=torch.uint8).cuda()
        out = torch.empty(text.size + 64, dtype=torch.uint8, device="cuda")
        ctx.enable_timing(True)
        try:
            m = ctx.decode_body_device(cb, buf[: (end_bit + 7) // 8], text.size, out, 3)
            t = ctx.timings("decode")
        finally:
            ctx.enable_timing(False)
        assert t["fixed_sync"] == (not missing)
        assert m == text.size and out[:m].cpu().numpy().tobytes() == text.tobytes(), missing


def test_large_stream(ctx):
    """64 MiB of four symbols (a DNA-like text): 16 Mi subsequences' worth of arithmetic, compared on the device."""
    import torch

    import entreepy_amd as E

    n = 64 << 20
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    text = (torch.randint(0, 4, (n,), generator=g, device="cuda", dtype=torch.int16) * 3 + 65).to(torch.uint8)
    enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device="cuda")
    dec = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    ctx.use_torch_stream()
    ln = ctx.encode_device(text, enc)
    ctx.enable_timing(True)
    try:
        m = ctx.decode_device(enc[4:ln], dec)
        t = ctx.timings("decode")
    finally:
        ctx.enable_timing(False)
    assert t["fixed_sync"] and m == n and torch.equal(dec[:n], text)


def test_eight_bit_codes_for_all_256_byte_values(ctx):
    """L = 8: a dictionary no encoder of the reference makes (it drops one of 256 symbols, Q1) -- 256 codewords of 8 bits in a
    random order; the decode is a byte permutation of the body."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(8)
    data_t = rng.permutation(256).astype(np.uint32)
    cb = E.Codebook.from_tables(data_t, np.full(256, 8, dtype=np.uint8))
    text = rng.integers(0, 256, size=120_001).astype(np.uint8)
    for start_bit in (0, 5):
        body, end_bit = O.pack_body(cb.data, cb.length, text, start_bit)
        buf = torch.frombuffer(bytearray(body) + bytearray(64), dtype=torch.uint8).cuda()
        out = torch.empty(text.size + 64, dtype=torch.uint8, device="cuda")
        ctx.enable_timing(True)
        try:
            m = ctx.decode_body_device(cb, buf[: (end_bit + 7) // 8], text.size, out, start_bit)
            t = ctx.timings("decode")
        finally:
            ctx.enable_timing(False)
        assert t["fixed_sync"] and m == text.size and out[:m].cpu().numpy().tobytes() == text.tobytes(), start_bit


@pytest.mark.parametrize("switch", ["ET_NO_FIXED_SYNC", "ET_NO_FIXED_WRITE"])
def test_fallbacks_still_decode_these_streams(switch):
    """ET_NO_FIXED_SYNC=1 (a child process: the switches are read once): the same streams through the exit maps;
    ET_NO_FIXED_WRITE=1: k_fixed_sync with the chained-table write behind it (what its start / count words are for)."""
    code = (
        "import numpy as np, entreepy_amd as E\n"
        "from tests.test_gpu_rowsync import flat\n"
        "from oracle import oracle as O\n"
        "c = E.Context(0); c.enable_timing(True)\n"
        "for k in (2, 4, 16, 64):\n"
        "    d = flat(k, 200_003 if k > 2 else 3_000_001, k)  # (1-bit codes: blocks of 65 536 symbols, four windows of the chained write's stage each)\n"
        "    et = O.encode(d.tobytes())\n"
        "    assert c.decode(et[4:]) == d.tobytes()\n"
        "    t = c.timings('decode')\n"
        "    assert t['fixed_sync'] == (SWITCH == 'ET_NO_FIXED_WRITE'), t\n"
        "    assert t['exhaustive_sync'] == (k != 2 or t['fixed_sync']), t  # (two 1-bit codewords without the arithmetic: the tree walk)\n"
        "    for cut in (1, 2, 9):\n"
        "        assert c.decode(et[4:-cut]) == O.decode(et[4:-cut])\n"
        "print('ok')\n"
    ).replace("SWITCH", repr(switch))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, **{switch: "1"}), timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]

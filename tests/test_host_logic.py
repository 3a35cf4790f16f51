"""CPU: the product's host half (et_codebook.cpp through the C ABI) against the
oracle, and the shape of the C ABI itself.  No GPU compute is called."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import entreepy_amd as E
from entreepy_amd import _native as N
from oracle import oracle as O
from tests.conftest import ROOT


def _same_tables(hist, text_len):
    try:
        od, ol, oo = O.build_dict(hist)
    except O.OracleError:
        with pytest.raises(E.EmptyInputError):
            E.Codebook.from_histogram(hist)
        return
    cb = E.Codebook.from_histogram(hist)
    assert (cb.data == od).all() and (cb.length == ol).all()
    assert cb.dfs_order.tolist() == oo.tolist()
    assert cb.header(text_len) == O.write_header(od, ol, text_len)
    assert cb.bits(hist) == int((hist * ol.astype(np.uint64)).sum())


def test_header_declares_what_the_library_exports():
    """Every function include/entreepy_hip.h declares is exported, and the binding
    declares the same set."""
    with open(os.path.join(ROOT, "include", "entreepy_hip.h")) as f:
        declared = set(re.findall(r"\b(et_[a-z0-9_]+)\s*\(", f.read()))
    declared -= {"et_status"}
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", N.LIB_PATH], text=True)
    exported = set(re.findall(r" T (et_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    lib = N.lib()
    assert b"gfx950" in lib.et_version()
    assert lib.et_encode_bound(100) == 7312  # 7200 + n (encode.zig:253-254), rounded to 16


def test_reference_fixtures_tables(res_files):
    for text in res_files.values():
        _same_tables(O.histogram(text), len(text))


@pytest.mark.parametrize("mode", range(6))
def test_random_histograms(mode):
    rng = np.random.default_rng(100 + mode)
    for _ in range(400):
        k = int(rng.integers(0, 257))
        hist = np.zeros(256, dtype=np.uint64)
        idx = rng.choice(256, size=k, replace=False)
        if mode == 0:
            vals = rng.integers(1, 4, size=k)
        elif mode == 1:
            vals = rng.integers(1, 1000, size=k)
        elif mode == 2:
            vals = 2 ** rng.integers(0, 40, size=k)
        elif mode == 3:
            vals = np.ones(k)
        elif mode == 4:
            vals = rng.geometric(0.001, size=k)
        else:
            vals = rng.integers(1, 3, size=k) * 1000  # many ties
        hist[idx] = np.asarray(vals, dtype=np.uint64)
        _same_tables(hist, int(hist.sum() % (1 << 40)))


def test_quirk_tables():
    hist = np.arange(1, 257, dtype=np.uint64)
    _same_tables(hist, 5)  # 256 distinct (Q1)
    assert E.Codebook.from_histogram(hist).raw.n_coded == 255
    one = np.zeros(256, dtype=np.uint64)
    one[97] = 4
    _same_tables(one, 4)  # single symbol (Q2)
    assert E.Codebook.from_histogram(one).header(4).hex() == "e7c0de010000000004"
    fib = [1, 1]
    while len(fib) < 45:
        fib.append(fib[-1] + fib[-2])
    h = np.zeros(256, dtype=np.uint64)
    h[10:55] = fib
    _same_tables(h, int(h.sum()))  # code lengths up to 44 (Q3)
    assert E.Codebook.from_histogram(h).raw.max_length == 44
    _same_tables(O.histogram(b"abc"), (1 << 34) + 5)  # 32-bit length wrap (Q4)


def test_parse_header_inverts_write_header(res_files):
    for text in list(res_files.values()) + [bytes(range(1, 256)) * 2, b"ab", bytes([0, 1, 0, 2, 0, 0, 3])]:
        et = O.encode(text)
        cb, n, off = E.parse_header(et[4:])
        od, ol, _ = O.build_dict(O.histogram(text))
        assert n == len(text)
        assert (cb.length == ol).all() and (cb.data[ol > 0] == od[ol > 0]).all()
        assert off + 4 == len(O.write_header(od, ol, len(text)))


def test_parse_header_rejects_malformed():
    et = O.encode(b"hello world, hello huffman")
    good = et[4:]
    with pytest.raises(E.EntreepyError):
        E.parse_header(good[:3])  # shorter than the fixed header
    with pytest.raises(E.EntreepyError):
        E.parse_header(good[:9])  # dictionary truncated
    # two entries with the same code -> not prefix-free
    bad = bytearray(good)
    cb, n, off = E.parse_header(good)
    dup = bytes([1]) + good[1:5] + bytes([ord("a"), 1, 0b00000000 | (ord("b") >> 1), ((ord("b") & 1) << 7) | 0, 0b10000000])
    with pytest.raises(E.EntreepyError):
        E.parse_header(dup)


def test_single_symbol_stream_parses_as_empty():
    cb, n, off = E.parse_header(bytes.fromhex("e7c0de010000000004")[4:])
    assert n == 4 and off == 5 and cb.raw.n_coded == 0


def test_gpu_calls_fail_loudly_without_a_device():
    """No CPU fallback: with no GPU in the process every compute entry point reports
    ET_ERR_HIP instead of computing something."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(E.EntreepyError) as ei:
        E.Context(0)
    assert ei.value.status == N.ET_ERR_HIP
    with pytest.raises(E.EntreepyError):
        E.encode(b"abc")


def test_cli_surface_without_gpu(tmp_path):
    """main.zig:45-67 help text and option errors do not need a device."""
    exe = os.path.join(ROOT, "entreepy_amd", "entreepy")
    out = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("Entreepy - Text compression tool\n\nUsage: entreepy [options] [command] [file] [command options]\n")
    assert "    -o, --output    output file (default: [file].et or decoded_[file])\n" in out.stdout
    assert subprocess.run([exe], capture_output=True, text=True).stdout == out.stdout  # main.zig:148-152
    assert subprocess.run([exe, "--help"], capture_output=True, text=True).stdout == out.stdout
    bad = subprocess.run([exe, "-x"], capture_output=True, text=True)
    assert bad.returncode != 0 and "invalid option: -x" in bad.stderr  # main.zig:116-118
    bad = subprocess.run([exe, "zip", "f"], capture_output=True, text=True)
    assert bad.returncode != 0 and "invalid command: zip" in bad.stderr  # main.zig:131-134


def test_parse_header_reports_codes_longer_than_32_bits():
    """Fibonacci counts give code lengths up to 44: the encoder reproduces the reference's
    u32 truncation (Q3), the decoder refuses such a stream (the reference would index out
    of bounds, Q9) with ET_ERR_UNSUPPORTED."""
    fib = [1, 1]
    while len(fib) < 45:
        fib.append(fib[-1] + fib[-2])
    h = np.zeros(256, dtype=np.uint64)
    h[10:55] = fib
    cb = E.Codebook.from_histogram(h)
    assert cb.raw.max_length == 44
    header = cb.header(int(h.sum()))
    with pytest.raises(E.EntreepyError) as ei:
        E.parse_header(header[4:])
    assert ei.value.status == N.ET_ERR_UNSUPPORTED


def test_check_magic():
    """The four bytes main.zig:204 strips unchecked (encode.zig:262-266 writes e7 c0 de 01)."""
    import ctypes

    from entreepy_amd import _native as N

    why = ctypes.c_char_p()
    assert N.lib().et_check_magic(bytes.fromhex("e7c0de01"), ctypes.byref(why)) == N.ET_OK and why.value is None
    assert N.lib().et_check_magic(bytes.fromhex("e7c0de02"), ctypes.byref(why)) == N.ET_ERR_FORMAT and b"version" in why.value
    assert N.lib().et_check_magic(b"PK\x03\x04", ctypes.byref(why)) == N.ET_ERR_FORMAT and b"magic" in why.value
    assert N.lib().et_check_magic(bytes.fromhex("e7c0de01"), None) == N.ET_OK


def test_cold_decode_block_plan_keeps_16_bytes_behind_every_inner_range():
    """sharded.cut_blocks: a last block shorter than 16 bytes is not cut off on its own (the range
    before it would have no run-out room, et_decode_range_sync refuses that)."""
    from entreepy_amd.sharded import cut_blocks

    assert cut_blocks(8192 * 5) == 5
    assert cut_blocks(8192 * 5 + 15) == 5
    assert cut_blocks(8192 * 5 + 16) == 6
    assert cut_blocks(7) == 1 and cut_blocks(8192) == 1 and cut_blocks(8192 + 1) == 1
    for nbytes in (8192 * 8 + 5, 8192 * 3 + 16, 100, 8192 * 2):
        n_blocks = cut_blocks(nbytes)
        for world in (1, 2, 3, 8, 16):
            for r in range(world):
                lo, hi = r * n_blocks // world, (r + 1) * n_blocks // world
                if hi > lo:
                    end = nbytes if hi == n_blocks else hi * 8192
                    assert end == nbytes or nbytes - end >= 16


def test_prefix_collision_check_follows_the_reference_loop():
    """et_prefix_collisions = encode.zig:221-247, including its k = 0 round (the bit above each code) and the
    u5 truncation of the bit index: compared with the loop restated here on crafted tables."""
    import ctypes

    from entreepy_amd import _native as N
    from entreepy_amd.codec import Codebook

    def reference_loop(data, length):
        out = []
        for i in range(256):
            for j in range(256):
                if length[i] == 0 or length[j] == 0 or i == j:
                    continue
                shorter = min(int(length[i]), int(length[j]))
                if all(((int(data[i]) >> ((int(length[i]) - k) & 31)) & 1) == ((int(data[j]) >> ((int(length[j]) - k) & 31)) & 1) for k in range(shorter + 1)):
                    out.append((i, j))
        return out

    rng = np.random.default_rng(11)
    tables = []
    d, l = np.zeros(256, dtype=np.uint32), np.zeros(256, dtype=np.uint8)
    d[65], l[65], d[66], l[66], d[67], l[67] = 0b1, 1, 0b11, 2, 0b0, 1  # "1" is a prefix of "11"
    tables.append((d, l))
    for _ in range(20):  # random tables, lengths up to 40 (the truncated index matters above 31)
        d, l = np.zeros(256, dtype=np.uint32), np.zeros(256, dtype=np.uint8)
        for s in rng.choice(256, size=12, replace=False):
            l[s] = rng.integers(1, 41)
            d[s] = rng.integers(0, 1 << 32, dtype=np.uint64) & ((1 << min(int(l[s]), 32)) - 1)
        tables.append((d, l))
    hits = 0
    for d, l in tables:
        cb = Codebook.from_tables(d, l)
        pairs = np.zeros(2 * 256 * 255, dtype=np.uint8)
        n = ctypes.c_size_t(0)
        assert N.lib().et_prefix_collisions(ctypes.byref(cb.raw), pairs.ctypes.data, pairs.size // 2, ctypes.byref(n)) == 0
        got = [(int(pairs[2 * k]), int(pairs[2 * k + 1])) for k in range(n.value)]
        assert got == reference_loop(d, l)
        hits += len(got)
    assert (65, 66) in reference_loop(*tables[0]) and hits > 2


def _brute_treewalk_table(data, length):
    """(row, byte) -> next row | completions << 8 | bit of the first << 12, from the codes alone: rows are the
    proper prefixes of the codes, numbered in the order the library meets them (symbols ascending, bits from
    the first), then 7 entry rows "skip b bits, then from the root"."""
    rows = {(): 0}
    codes = {}
    for s in range(256):
        if length[s]:
            bits = tuple((int(data[s]) >> i) & 1 for i in range(int(length[s]) - 1, -1, -1))
            codes[bits] = s
            for k in range(1, len(bits)):
                rows.setdefault(bits[:k], len(rows))
    n_int = len(rows)
    prefix_of = {v: k for k, v in rows.items()}
    table = np.zeros((n_int + 7) * 256, dtype=np.uint16)
    for r in range(n_int + 7):
        for f in range(256):
            cur = prefix_of[r] if r < n_int else ()
            skip = 0 if r < n_int else r - n_int + 1
            n = first = 0
            for i in range(skip, 8):
                cur = cur + ((f >> (7 - i)) & 1,)
                if cur in codes:
                    if n == 0:
                        first = i
                    n += 1
                    cur = ()
            table[r * 256 + f] = rows[cur] | (n << 8) | (first << 12)
    return n_int, table


def test_treewalk_table_against_brute_force():
    """et_treewalk_table (tw_build_tree + tw_fill_table, the reference the device-built table is compared with on
    the GPU) equals a table derived from the codes by brute force, for code tables of the product's own
    construction; dictionaries that are not full trees are turned away."""
    import ctypes

    from entreepy_amd import _native as N
    from entreepy_amd.codec import Codebook

    rng = np.random.default_rng(5)
    for trial in range(12):
        k = int(rng.integers(2, 257)) if trial else 256
        h = np.zeros(256, dtype=np.uint64)
        h[rng.choice(256, size=k, replace=False)] = rng.integers(1, 1 << int(rng.integers(2, 30)), size=k)
        cb = Codebook.from_histogram(h)
        if cb.raw.max_length > 32:
            continue
        n_int = ctypes.c_uint32(0)
        cap = (256 + 7) * 256
        table = np.zeros(cap, dtype=np.uint16)
        rc = N.lib().et_treewalk_table(ctypes.byref(cb.raw), table.ctypes.data, cap, ctypes.byref(n_int))
        assert rc == N.ET_OK
        want_n, want = _brute_treewalk_table(cb.data, cb.length)
        assert n_int.value == want_n == cb.raw.n_coded - 1
        assert np.array_equal(table[: want.size], want)
    # one symbol: no tree; a prefix-free set that is not a full tree: not for the walk
    one = Codebook.from_tables(np.zeros(256, dtype=np.uint32), np.zeros(256, dtype=np.uint8))
    n_int = ctypes.c_uint32(0)
    assert N.lib().et_treewalk_table(ctypes.byref(one.raw), None, 0, ctypes.byref(n_int)) == N.ET_ERR_UNSUPPORTED
    d, l = np.zeros(256, dtype=np.uint32), np.zeros(256, dtype=np.uint8)
    d[65], l[65], d[66], l[66] = 0b0, 1, 0b10, 2  # "11" is missing
    gap = Codebook.from_tables(d, l)
    assert N.lib().et_treewalk_table(ctypes.byref(gap.raw), None, 0, ctypes.byref(n_int)) == N.ET_ERR_UNSUPPORTED


def _chain_tables(cb):
    import ctypes

    from entreepy_amd import _native as N

    n_entries, n_tables = ctypes.c_uint32(0), ctypes.c_uint32(0)
    table = np.zeros(4096, dtype=np.uint64)
    first = np.zeros(256, dtype=np.uint32)
    bits = np.zeros(256, dtype=np.uint8)
    rc = N.lib().et_chain_tables(ctypes.byref(cb.raw), table.ctypes.data, table.size, ctypes.byref(n_entries), first.ctypes.data, bits.ctypes.data, 256,
                                 ctypes.byref(n_tables))
    assert rc == N.ET_OK
    return table[: n_entries.value], first[: n_tables.value], bits[: n_tables.value]


def _decode_through_chain(table, stream_bits, n_symbols):
    """Greedy multi-symbol decode of a bit list through the chained tables, following every entry's own "next table"
    and "next shift" fields the way the write walk does (window = the next 32 bits, zero-padded)."""
    out = []
    pos, t_off, shift = 0, 0, 32 - 11
    padded = stream_bits + [0] * 64
    while len(out) < n_symbols:
        window = 0
        for b in padded[pos : pos + 32]:
            window = (window << 1) | b
        e = int(table[t_off // 8 + (window >> shift)])
        lo, hi = e & 0xFFFFFFFF, e >> 32
        adv = lo & 0xFFFF
        if adv & 0x8000:
            adv -= 0x10000
        n = (adv + 15) >> 10
        used = (n << 10) - adv
        assert 1 <= used <= 11 and 0 <= n <= 2
        if n >= 1:
            out.append((lo >> 16) & 0xFF)
            assert 1 <= (lo >> 24) <= used
        else:
            assert (lo >> 24) == 0
        if n == 2:
            out.append((hi >> 16) & 0xFF)
        pos += used
        t_off, shift = hi & 0xFFFF, hi >> 24
        if n >= 1:
            assert t_off == 0 and shift == 32 - 11
    return out[:n_symbols], pos


def test_chain_tables_decode_what_the_codes_say():
    """The chained lookup tables of the write walk (et_chain_tables: tw_chain_plan + tw_chain_entry, the same code the
    device fill runs): decoding random symbol strings through them gives the strings back, for short, deep (Fibonacci
    weights: codes up to 32 bits), flat, two-symbol and enwik-like code tables; the plan respects its bounds."""
    from entreepy_amd.codec import Codebook

    from . import corpus

    rng = np.random.default_rng(11)
    hists = []
    for k in (2, 3, 17, 93, 256):
        h = np.zeros(256, dtype=np.uint64)
        h[rng.choice(256, size=k, replace=False)] = rng.integers(1, 1 << 20, size=k)
        hists.append(h)
    fib = np.zeros(256, dtype=np.uint64)
    a, b = 1, 1
    for s in range(33):  # 33 Fibonacci weights: a 32-level caterpillar
        fib[s + 40] = a
        a, b = b, a + b
    hists.append(fib)
    hists.append(np.ones(256, dtype=np.uint64))  # flat 8-bit codes
    hists.append((corpus.enwik_like_distribution() * (1 << 40)).astype(np.uint64))
    hists.append((corpus.midsummer_distribution() * (1 << 30)).astype(np.uint64))
    for h in hists:
        cb = Codebook.from_histogram(h)
        assert cb.raw.max_length <= 32
        table, first, bits = _chain_tables(cb)
        assert first[0] == 0 and bits[0] == 11 and table.size == int((1 << bits.astype(np.uint32)).sum())
        assert table.size <= 2048 + 576 and first.size <= 256
        coded = np.flatnonzero(cb.length)
        # every symbol appears, the long ones often: uniform over the coded symbols
        text = rng.choice(coded, size=3000)
        stream = []
        for s in text:
            L = int(cb.length[s])
            stream += [(int(cb.data[s]) >> i) & 1 for i in range(L - 1, -1, -1)]
        got, pos = _decode_through_chain(table, stream, text.size)
        assert got == [int(s) for s in text]


def test_row_code_is_the_flat_codes_of_the_reference_builder_only():
    """et_row_code (csrc/et_rowsync_host.cpp): which dictionaries a decode synchronises by rows and columns -- complete codes
    of 7 and 8 bits whose 7-bit codewords are the values 0 .. t-1.  The reference's builder gives exactly that for k = 129 .. 255
    symbols of equal weight (t = 256 - k); anything else goes the general way."""
    import ctypes

    import entreepy_amd as E
    from entreepy_amd import _native as N

    def row_t(cb):
        t = ctypes.c_uint32(0xdead)
        rc = N.lib().et_row_code(ctypes.byref(cb.raw), ctypes.byref(t))
        assert rc in (N.ET_OK, N.ET_ERR_UNSUPPORTED)
        return t.value if rc == N.ET_OK else None

    for k in (129, 130, 200, 254, 255):
        h = np.zeros(256, dtype=np.uint64)
        h[:k] = 1000
        cb = E.Codebook.from_histogram(h)
        assert row_t(cb) == 256 - k, k
        # the same code with its 7-bit codewords elsewhere (bitwise complement of every codeword: still complete and
        # prefix-free, but they are now the LAST values): not a row code
        length, data = cb.length, cb.data
        flipped = np.where(length > 0, (~data) & ((1 << length.astype(np.uint32)) - 1), 0).astype(np.uint32)
        assert row_t(E.Codebook.from_tables(flipped, length)) is None, k
    # 256 symbols of 8 bits each (a hand-made dictionary: the reference's own encoder drops one of 256): t = 0
    full = E.Codebook.from_tables(np.arange(256, dtype=np.uint32), np.full(256, 8, dtype=np.uint8))
    assert row_t(full) == 0
    # not complete / other lengths / text
    gap = E.Codebook.from_tables(np.arange(256, dtype=np.uint32), np.where(np.arange(256) < 255, 8, 0).astype(np.uint8))
    assert row_t(gap) is None
    h = np.zeros(256, dtype=np.uint64)
    h[:100] = 7
    assert row_t(E.Codebook.from_histogram(h)) is None  # 6 and 7 bits
    from tests import corpus

    assert row_t(E.Codebook.from_histogram(np.bincount(corpus.text_like(100_000, 1), minlength=256))) is None


def test_decode_path_of_flat_alphabets():
    """et_decode_path (diagnostics): which synchronisation a one-GPU decode starts with, from the code table alone.  k symbols of
    equal weight through the reference's builder: powers of two are fixed-length codes; codes of L and L + 1 bits take the tree
    walk when et::quick_to_synchronise expects them to settle -- the table below is what the GPU runs measured
    (profiles/r04_flat_alphabets.jsonl: the tree walk settles the 'T' rows, most blocks give up on the others) -- and the exit
    maps or, with 7 and 8 bits, the row walk when not; text takes the tree walk."""
    import ctypes

    import entreepy_amd as E
    from entreepy_amd import _native as N
    from tests import corpus

    def path(cb):
        p = ctypes.c_uint32(99)
        assert N.lib().et_decode_path(ctypes.byref(cb.raw), ctypes.byref(p)) == N.ET_OK
        return "TMRFW"[p.value]

    def flat_code(k):
        h = np.zeros(256, dtype=np.uint64)
        h[:k] = 1000
        return E.Codebook.from_histogram(h)

    want = {2: "F", 3: "T", 4: "F", 5: "T", 7: "T", 8: "F", 10: "T", 14: "T", 16: "F", 17: "T", 26: "T", 30: "T", 31: "M", 32: "F", 33: "M", 34: "T", 36: "T", 50: "T", 58: "T",
            60: "T", 61: "M", 62: "M", 64: "F", 65: "M", 68: "T", 72: "T", 100: "T", 112: "T", 120: "M", 124: "M", 128: "F", 129: "R", 136: "R", 150: "T", 160: "T",
            200: "T", 205: "T", 215: "R", 240: "R", 254: "R", 255: "R"}
    got = {k: path(flat_code(k)) for k in want}
    assert got == want, {k: (got[k], want[k]) for k in want if got[k] != want[k]}
    assert path(E.Codebook.from_histogram(np.bincount(corpus.text_like(100_000, 1), minlength=256))) == "T"
    two = np.zeros(256, dtype=np.uint64)
    two[65], two[66] = 1000, 3
    assert path(E.Codebook.from_histogram(two)) == "F"  # two codewords of one bit, however skewed
    # hand-made: 256 codewords of 8 bits (fixed), the same less one (not complete: not fixed, and no row code either)
    assert path(E.Codebook.from_tables(np.arange(256, dtype=np.uint32), np.full(256, 8, dtype=np.uint8))) == "F"
    assert path(E.Codebook.from_tables(np.arange(256, dtype=np.uint32), np.where(np.arange(256) < 255, 8, 0).astype(np.uint8))) == "M"


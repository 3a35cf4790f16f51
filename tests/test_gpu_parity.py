"""GPU parity: the HIP path (through the C ABI) against the oracle, bit-exact."""
import hashlib

import numpy as np
import pytest

from tests import corpus

pytestmark = pytest.mark.gpu

GOLDEN_SHA = {
    "test.txt": "761b5bc3dcb9d8487eaa764b7c1b207774caff78a464b229c56f939417af764d",
    "nice.shakespeare.txt": "795f27fd81733435fbaa1e58d260950eec57f800652245464b7e3209407a2409",
    "a_midsummer_nights_dream.txt": "d152197f8c5ee87c68ca812929ebc92ffd74b0fecbdd6c491b203d19479eb97b",
}


def _oracle():
    from oracle import oracle as O

    return O


def _roundtrip(ctx, data, expect_lossless=True):
    O = _oracle()
    data = bytes(data)
    want = O.encode(data)
    got = ctx.encode(data)
    assert got == want, f"encode mismatch n={len(data)} first diff at {next((i for i,(a,b) in enumerate(zip(got,want)) if a!=b), min(len(got),len(want)))} len {len(got)} vs {len(want)}"
    back = ctx.decode(got[4:])
    assert back == O.decode(want[4:])
    if expect_lossless and len(set(data)) > 1:  # a lone symbol encodes to the bare header (Q2)
        assert back == data
    return got


@pytest.mark.parametrize("name", list(GOLDEN_SHA))
def test_reference_fixtures(ctx, res_files, name):
    """src/test.zig:35-72 round trips, plus the golden .et bytes."""
    et = _roundtrip(ctx, res_files[name])
    assert hashlib.sha256(et).hexdigest() == GOLDEN_SHA[name]
    if name == "nice.shakespeare.txt":
        assert len(et) == 374  # README.md:51


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 31, 33, 255, 256, 257, 4095, 4096, 4097, 8191, 65535, 65536, 65537, 262144 + 5, 1 << 20, (1 << 22) + 123])
def test_text_like_sizes(ctx, n):
    _roundtrip(ctx, corpus.text_like(n, seed=n))


def test_empty_input(ctx):
    import entreepy_amd as E

    with pytest.raises(E.EmptyInputError):
        ctx.encode(b"")


def test_single_symbol(ctx):
    et = _roundtrip(ctx, b"a" * 1000, expect_lossless=False)
    assert et == bytes.fromhex("e7c0de0100000003e8")
    assert ctx.decode(et[4:]) == b""


@pytest.mark.parametrize("n", [2, 100, 70000])
def test_two_symbols(ctx, n):
    _roundtrip(ctx, corpus.uniform(n, 3, 65, 67) if n > 2 else b"ab")


@pytest.mark.parametrize("n", [5000, 300000, (1 << 21) + 17])
def test_uniform_255(ctx, n):
    """Bytes 1..255: the largest alphabet the reference encodes losslessly."""
    _roundtrip(ctx, corpus.uniform(n, 5, 1, 256))


def test_nul_bytes(ctx):
    """NUL symbols hang the reference decoder (Q6); ours must handle them."""
    _roundtrip(ctx, corpus.uniform(100000, 6, 0, 40))


def test_uniform_256_quirk(ctx):
    """All 256 byte values: the most frequent symbol gets no code (Q1).  Encode is
    bit-exact with the reference semantics; the stream is lossy by construction."""
    _roundtrip(ctx, corpus.uniform(400000, 7, 0, 256), expect_lossless=False)


def test_tail_symbols(ctx, res_files):
    """Midsummer + 'eee': the reference decoder drops 3 symbols (Q7); ours does not."""
    _roundtrip(ctx, res_files["a_midsummer_nights_dream.txt"] + b"eee")


def test_histogram_device(ctx):
    import torch

    data = corpus.text_like(1 << 20, 11)
    t = torch.from_numpy(data).cuda()
    for off, n in [(0, data.size), (1, 1000), (7, 65536), (13, (1 << 20) - 13), (16, 4096), (5, 1)]:
        h = torch.zeros(256, dtype=torch.int64, device="cuda")
        ctx.histogram_device(t[off : off + n], h)
        torch.cuda.synchronize()
        want = np.bincount(data[off : off + n], minlength=256)
        assert (h.cpu().numpy() == want).all(), (off, n)


def test_device_entry_points_misaligned_input(ctx):
    import torch

    import entreepy_amd as E

    O = _oracle()
    data = corpus.text_like(300000, 12)
    t = torch.from_numpy(data).cuda()
    for off in (0, 1, 3, 8, 15):
        view = t[off:]
        out = torch.zeros(E.encode_bound(view.numel()), dtype=torch.uint8, device="cuda")
        n = ctx.encode_device(view, out)
        torch.cuda.synchronize()
        et = out[:n].cpu().numpy().tobytes()
        assert et == O.encode(data[off:]), off
        # decode with the compressed stream at the same odd offsets
        comp = torch.zeros(n + 32, dtype=torch.uint8, device="cuda")
        comp[off : off + n - 4] = out[4:n]
        dec = torch.empty(view.numel() + 64, dtype=torch.uint8, device="cuda")
        m = ctx.decode_device(comp[off : off + n - 4], dec)
        torch.cuda.synchronize()
        assert m == view.numel() and (dec[:m] == view).all(), off


def _random_prefix_code(rng, n_sym, max_len):
    """A prefix-free table with random lengths up to max_len (Kraft-feasible), codes
    assigned canonically; not optimal, only a legal input for the body kernels."""
    lens = np.sort(rng.integers(2, max_len + 1, size=n_sym))
    while sum(2.0 ** -int(l) for l in lens) > 1.0:
        lens[np.argmin(lens)] += 1
        lens = np.sort(lens)
    code, prev, codes = 0, int(lens[0]), []
    for l in lens:
        code <<= int(l) - prev
        prev = int(l)
        codes.append(code)
        code += 1
    return lens, codes


@pytest.mark.parametrize("max_len,seed", [(12, 1), (16, 2), (24, 3), (32, 4)])
def test_body_random_tables_and_start_bits(ctx, max_len, seed):
    """et_encode_body_device / et_decode_body_device with arbitrary (non-Huffman)
    prefix codes and every start-bit phase, against oracle pack_body."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(seed)
    n_sym = 60
    lens, codes = _random_prefix_code(rng, n_sym, max_len)
    syms = rng.choice(256, size=n_sym, replace=False)
    data_t, len_t = np.zeros(256, np.uint32), np.zeros(256, np.uint8)
    for s, l, c in zip(syms, lens, codes):
        data_t[s], len_t[s] = c & 0xFFFFFFFF, l
    cb = E.Codebook.from_tables(data_t, len_t)
    text = syms[rng.integers(0, n_sym, size=150000)].astype(np.uint8)
    t = torch.from_numpy(text).cuda()
    h = torch.zeros(256, dtype=torch.int64, device="cuda")
    for start_bit in (0, 1, 7, 31, 32, 45, 8 * 13 + 3):
        ctx.histogram_device(t, h)
        out = torch.full((text.size * 4 + 64,), 0xFF, dtype=torch.uint8, device="cuda")
        end = ctx.encode_body_device(cb, t, out, start_bit)
        torch.cuda.synchronize()
        want, want_end = O.pack_body(data_t, len_t, text, start_bit)
        assert end == want_end
        got = out[: (end + 7) // 8].cpu().numpy()
        first_word, last_word = start_bit // 32, (end - 1) // 32
        w = np.frombuffer(want, dtype=np.uint8)
        assert (got[first_word * 4 :] == w[first_word * 4 :]).all(), start_bit
        # decode from the same buffer (start_bit < 8 only: API contract)
        if start_bit < 8:
            dec = torch.empty(text.size + 64, dtype=torch.uint8, device="cuda")
            m = ctx.decode_body_device(cb, out[: (end + 7) // 8], text.size, dec, start_bit)
            torch.cuda.synchronize()
            assert m == text.size and (dec[:m].cpu().numpy() == text).all()


def test_long_codes_beyond_32_bits(ctx):
    """Code lengths above 32: the reference emits bit (data >> ((j-1) & 31)) & 1
    (encode.zig:311), deterministic garbage that must still be bit-exact (Q3)."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(9)
    data_t, len_t = np.zeros(256, np.uint32), np.zeros(256, np.uint8)
    for s in range(40):
        data_t[s] = rng.integers(0, 1 << 32, dtype=np.uint64)
        len_t[s] = [1, 5, 31, 32, 33, 40, 64, 65, 100, 255][s % 10]
    cb = E.Codebook.from_tables(data_t, len_t)
    text = rng.integers(0, 40, size=70001, dtype=np.uint8)
    t = torch.from_numpy(text).cuda()
    h = torch.zeros(256, dtype=torch.int64, device="cuda")
    for start_bit in (0, 5, 37):
        ctx.histogram_device(t, h)
        want, want_end = O.pack_body(data_t, len_t, text, start_bit)
        out = torch.zeros(len(want) + 64, dtype=torch.uint8, device="cuda")
        end = ctx.encode_body_device(cb, t, out, start_bit)
        torch.cuda.synchronize()
        assert end == want_end
        assert out[: len(want)].cpu().numpy().tobytes()[start_bit // 32 * 4 :] == want[start_bit // 32 * 4 :]


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_virtual_shards_concat(ctx, shards):
    """Multi-GPU encode logic on one device: per-shard histograms summed, one code
    table, per-shard bodies at their global bit offsets, boundary bytes OR-merged."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    data = corpus.text_like(1_000_003, 21)
    want = O.encode(data)
    t = torch.from_numpy(data).cuda()
    bounds = np.linspace(0, data.size, shards + 1).astype(np.int64)
    hists = []
    for r in range(shards):
        h = torch.zeros(256, dtype=torch.int64, device="cuda")
        ctx.histogram_device(t[bounds[r] : bounds[r + 1]], h)
        hists.append(h.cpu().numpy().astype(np.uint64))
    total = np.sum(hists, axis=0).astype(np.uint64)
    cb = E.Codebook.from_histogram(total)
    header = cb.header(data.size)
    file_img = np.zeros(len(want) + 8, dtype=np.uint8)
    file_img[: len(header)] = np.frombuffer(header, dtype=np.uint8)
    bit = len(header) * 8
    for r in range(shards):
        view = t[bounds[r] : bounds[r + 1]]
        h = torch.zeros(256, dtype=torch.int64, device="cuda")
        ctx.histogram_device(view, h)
        out = torch.zeros(view.numel() + 64, dtype=torch.uint8, device="cuda")
        local_start = bit % 32
        end = ctx.encode_body_device(cb, view, out, local_start)
        torch.cuda.synchronize()
        assert end - local_start == cb.bits(hists[r])
        piece = out[: (end + 7) // 8].cpu().numpy()
        base_byte = (bit // 32) * 4
        file_img[base_byte : base_byte + piece.size] |= piece
        bit += end - local_start
    assert file_img[: (bit + 7) // 8].tobytes() == want


@pytest.mark.parametrize("rounds", [1, 2, 4, 8, 16, 32, 64, 128])
@pytest.mark.parametrize("n", [70001, (3 << 20) + 77])
def test_every_tile_geometry(ctx, rounds, n):
    """Tiles of 1..128 rounds = 4 .. 512 KiB (the size-based choice reaches 128 only at 1 GiB)."""
    ctx.set_tile_rounds(rounds)
    try:
        _roundtrip(ctx, corpus.text_like(n, seed=rounds * 1000 + 7))
        _roundtrip(ctx, corpus.uniform(n // 3, rounds, 1, 256))
    finally:
        ctx.set_tile_rounds(0)


def test_grid_stride_and_multi_round_tiles(ctx):
    """48 MiB: more tiles than the 2048-workgroup grid and 2-round tiles by size."""
    import hashlib

    O = _oracle()
    data = corpus.text_like(48 << 20, 99)
    got = ctx.encode(data)
    want = O.encode(data)
    assert hashlib.sha256(got).digest() == hashlib.sha256(want).digest() and got == want
    assert ctx.decode(got[4:]) == data.tobytes()


@pytest.mark.parametrize("k", [3, 4, 5, 16, 17, 64, 100, 200])
def test_uniform_alphabets_fast_and_exhaustive_sync(ctx, k):
    """Uniform k-symbol streams: (nearly) fixed-length codes barely self-synchronise, so
    the decoder takes its exhaustive path (exit map per start + map composition); both
    paths must agree with the oracle."""
    n = 200_003
    data = corpus.uniform(n, 1000 + k, 1, 1 + k)
    _roundtrip(ctx, data)
    ctx.enable_timing(True)
    try:
        et = ctx.encode(data)
        assert ctx.decode(et[4:]) == data.tobytes()
    finally:
        ctx.enable_timing(False)


def test_skewed_then_flat_stream_hits_the_sync_cap(ctx):
    """A code with spread lengths (fast path chosen by the host) on a stream whose second
    half uses only the long, equal-length codes: many blocks exceed the local trip cap,
    and the decoder must still be exact (sweep-0 give-up -> exhaustive path or repairs)."""
    rng = np.random.default_rng(5)
    head = corpus.text_like(300_000, 8)
    rare = np.array([s for s in range(256) if s not in set(head.tolist())][:64], dtype=np.uint8)
    tail = rare[rng.integers(0, rare.size, size=400_000)]
    _roundtrip(ctx, np.concatenate([head, tail, head[:1000]]))


@pytest.mark.parametrize("p_short", [0.02, 0.1, 0.3])
def test_slowly_synchronising_streams(ctx, p_short):
    """Mostly equal-length codes with a sprinkling of short ones: a wrong start survives for
    many codewords, so lanes are re-walked several times and merge with their earlier walk
    late (k_dec_sync_reg's checkpointed re-walk, trip cap, repair sweeps)."""
    rng = np.random.default_rng(int(p_short * 1000))
    n = (3 << 20) + 1234
    flat = rng.integers(16, 16 + 160, size=n).astype(np.uint8)
    short = rng.integers(1, 4, size=n).astype(np.uint8)
    data = np.where(rng.random(n) < p_short, short, flat).astype(np.uint8)
    _roundtrip(ctx, data)


def test_property_random_streams(ctx):
    """hypothesis: arbitrary byte strings -- GPU encode == oracle encode, GPU decode ==
    oracle's intended decode (== input whenever the format is lossless)."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    O = _oracle()

    @settings(max_examples=150, deadline=None)
    @given(st.one_of(
        st.binary(min_size=1, max_size=3000),
        st.lists(st.sampled_from([0, 1, 2, 3, 255]), min_size=1, max_size=5000).map(bytes),
        st.tuples(st.integers(1, 255), st.integers(1, 70000)).map(lambda t: bytes([t[0]]) * t[1]),
    ))
    def check(data):
        want = O.encode(data)
        got = ctx.encode(data)
        assert got == want
        assert ctx.decode(got[4:]) == O.decode(want[4:])

    check()


def test_decode_rejects_malformed_streams(ctx):
    """The reference validates nothing (main.zig:199 TODO); the library must fail with a
    status, never crash or hang, on truncated or corrupted inputs."""
    import entreepy_amd as E

    O = _oracle()
    text = corpus.text_like(50_000, 3)
    et = O.encode(text)
    good = et[4:]
    for cut in (0, 3, 4, 8, 40):
        with pytest.raises(E.EntreepyError):
            ctx.decode(good[:cut])
    # body truncated: decodes what is there, fewer symbols than declared
    _, n, off = E.parse_header(good)
    part = ctx.decode(good[: off + 1000])
    assert 0 < len(part) < n and text.tobytes().startswith(part)
    # dictionary corrupted so that two codes collide
    bad = bytearray(good)
    bad[7] ^= 0xFF
    try:
        out = ctx.decode(bytes(bad))
        assert len(out) <= n  # a still-valid (different) code table: any bounded output is acceptable
    except E.EntreepyError:
        pass
    # random garbage after a valid header must terminate
    rng = np.random.default_rng(1)
    junk = good[:off] + rng.integers(0, 256, size=20000, dtype=np.uint8).tobytes()
    assert len(ctx.decode(junk)) <= n


def test_decode_fuzzed_bodies_match_the_oracle(ctx):
    """Bit flips, truncations and junk in the BODY of streams with intact headers, large
    enough (several 8 KiB blocks, ragged ends) to run through the register-window
    kernels: whatever the bits say, the GPU decode equals the oracle's intended decoder
    (symbols whose code BEGINS before the stream's end, at most the declared count)."""
    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(2024)
    for trial in range(12):
        n = int(rng.integers(40_000, 400_000))
        text = corpus.text_like(n, 500 + trial)
        good = bytearray(O.encode(text)[4:])
        _, _, off = E.parse_header(bytes(good))
        kind = trial % 4
        if kind == 0:  # scattered bit flips
            for pos in rng.integers(off, len(good), size=50):
                good[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:  # a run of junk in the middle
            a = int(rng.integers(off, len(good) - 5000))
            good[a : a + 4096] = rng.integers(0, 256, size=4096, dtype=np.uint8).tobytes()
        elif kind == 2:  # truncated at a ragged place
            del good[int(rng.integers(off + 1, len(good))) :]
        else:  # all-ones / all-zeros tails (longest and shortest codes back to back)
            good[-9000:-4500] = b"\xff" * 4500
            good[-4500:] = b"\x00" * 4500
        want = O.decode(bytes(good))
        assert ctx.decode(bytes(good)) == want, f"trial {trial} kind {kind}"


@pytest.mark.parametrize("where", ["front", "middle", "several"])
def test_runs_of_one_long_code_do_not_settle(ctx, where):
    """Long runs of 0xff / 0x00 inside a text stream decode as one long code repeated: a wrong
    start never re-synchronises there, blocks hit the trip cap (start marker 0xff), the
    speculative write must not act on that state (it once walked from the marker and
    overwrote its own LDS tables: an intermittent hang) and the result must still be the
    oracle's -- also when the run covers the stream's FIRST block, whose start the
    verification has to check against the known first bit."""
    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(11)
    text = corpus.text_like(700_000, 41)
    good = bytearray(O.encode(text)[4:])
    _, _, off = E.parse_header(bytes(good))
    spans = {"front": [(off, 30_000)], "middle": [(off + 150_000, 40_000)],
             "several": [(off + 20_000, 9_000), (off + 100_000, 25_000), (off + 300_000, 17_000)]}[where]
    for i, (a, ln) in enumerate(spans):
        good[a : a + ln] = (b"\xff" if i % 2 == 0 else b"\x00") * ln
    want = O.decode(bytes(good))
    for _ in range(6):  # the failure was timing dependent
        assert ctx.decode(bytes(good)) == want


@pytest.mark.parametrize("p_common", [0.9, 0.99, 0.999])
def test_heavily_skewed_streams(ctx, p_common):
    """One symbol dominates: 1-bit codes, up to 65536 symbols per 8 KiB block (the write
    kernel's multi-window path), and long codes for the rare symbols next to them."""
    rng = np.random.default_rng(int(p_common * 1000))
    n = 1_500_000
    data = np.full(n, ord("a"), dtype=np.uint8)
    rare = rng.random(n) > p_common
    data[rare] = rng.integers(0, 200, size=int(rare.sum()), dtype=np.uint8)
    _roundtrip(ctx, data)


@pytest.mark.parametrize("config", ["2-text-5M", "3-text-100M", "3-enwik-like-100M"])
def test_baseline_configs_bit_exact(ctx, config):
    """BASELINE.json configs 2 and 3 at their full sizes (the corpora do not exist
    offline, so Midsummer tiled to 5 458 199 B and 10^8 order-0 samples of its distribution,
    seed 0x5EED0003 -- or the real files when $ET_CORPUS_SHAKESPEARE / $ET_CORPUS_ENWIK8 name them):
    the GPU .et image is the oracle's, byte for byte, and decodes back.  "enwik-like": 10^8 bytes
    of the 206-symbol stream with code lengths up to 24 (the long codes enwik has and text has not).
    (Config 1 is test_reference_fixtures, config 4 test_gpu_configs.py::test_text_1gib_image_is_byte_exact
    and the bench, config 5 test_gpu_configs.py::test_config5_*.)"""
    O = _oracle()
    if config.startswith("2"):
        data = corpus.from_env("ET_CORPUS_SHAKESPEARE")
        data = corpus.tiled_midsummer(5_458_199) if data is None else data
    elif config == "3-text-100M":
        data = corpus.from_env("ET_CORPUS_ENWIK8")
        data = corpus.text_like(100_000_000, 0x5EED0003) if data is None else data
    else:
        data = corpus.enwik_like(100_000_000, 0x5EED0008)
        _, ol, _ = O.build_dict(O.histogram(data))
        assert int((ol > 0).sum()) >= 200 and int(ol.max()) >= 20
    want = O.encode(data)
    got = ctx.encode(data)
    assert len(got) == len(want) and hashlib.sha256(got).digest() == hashlib.sha256(want).digest()
    back = ctx.decode(got[4:])
    assert len(back) == data.size and hashlib.sha256(back).digest() == hashlib.sha256(data.tobytes()).digest()


def test_decode_fuzz_mixed_sources(ctx):
    """80 streams from three kinds of source (text, uniform alphabets of random size,
    text diluted with one dominant symbol), each hit by one to four of: a bit flip, a run of
    random bytes, a run of 0xff, a run of 0x00 (up to 9 KB, anywhere in the body), a
    truncation.  Every decode must return (no hang, no crash) and equal the oracle's."""
    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(777)
    for trial in range(80):
        n = int(rng.integers(30_000, 600_000))
        src = trial % 3
        if src == 0:
            text = corpus.text_like(n, 900 + trial)
        elif src == 1:
            text = corpus.uniform(n, 900 + trial, 1, 1 + int(rng.integers(2, 255)))
        else:
            p = float(rng.choice([0.5, 0.9, 0.99]))
            text = np.where(rng.random(n) < p, 32, corpus.text_like(n, 900 + trial)).astype(np.uint8)
        good = bytearray(O.encode(text)[4:])
        _, _, off = E.parse_header(bytes(good))
        if len(good) - off < 64:
            continue
        for _ in range(int(rng.integers(1, 5))):
            k, a, ln = int(rng.integers(0, 5)), int(rng.integers(off, len(good))), int(rng.integers(1, 9000))
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
        assert ctx.decode(bytes(good)) == O.decode(bytes(good)), f"trial {trial} source {src}"


def test_decode_fuzz_dictionaries(ctx):
    """Bit flips in the HEADER and dictionary (the reference parses them unchecked,
    decode.zig:61-141): the call returns an error or some bounded output -- code tables that
    are still prefix-free but incomplete make the walk skip bits where no code matches -- and
    never hangs or crashes, whatever the table looks like."""
    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(4242)
    accepted = 0
    for trial in range(120):
        n = int(rng.integers(20_000, 300_000))
        text = corpus.text_like(n, 1300 + trial) if trial % 2 else corpus.uniform(n, 1300 + trial, 1, 1 + int(rng.integers(2, 200)))
        good = bytearray(O.encode(text)[4:])
        _, declared, off = E.parse_header(bytes(good))
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, off))
            good[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            _, declared, _ = E.parse_header(bytes(good))
            out = ctx.decode(bytes(good))
        except E.EntreepyError:
            continue
        accepted += 1
        assert len(out) <= declared
    assert accepted > 5  # some corrupted tables do remain decodable


def test_two_contexts_on_two_threads_host_pointer_calls(ctx):
    """SURVEY 8b "Threading": the library is thread-safe per et_ctx handle and keeps no hidden
    globals -- two host threads, each with its own context (own stream, workspaces, staging),
    encode and decode different streams at the same time; every result equals the oracle's."""
    import threading

    import entreepy_amd as E

    O = _oracle()
    jobs = [[corpus.text_like(n, 900 + i).tobytes() for i, n in enumerate(sizes)]
            for sizes in ([3_000_001, 70_000, 1, 1_500_000], [2_000_003, 5, 900_000, 3_100_000])]
    jobs[1].append(corpus.uniform(400_000, 7, 1, 201).tobytes())  # the exhaustive path, beside the other thread's text
    want = [[O.encode(d) for d in js] for js in jobs]
    errors = []

    def work(k):
        try:
            mine = E.Context(0)
            for rep in range(3):
                for d, w in zip(jobs[k], want[k]):
                    et = mine.encode(d)
                    assert et == w, f"thread {k}: encode differs (n={len(d)})"
                    assert mine.decode(et[4:]) == O.decode(w[4:]), f"thread {k}: decode differs (n={len(d)})"
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker thread hung"
    assert not errors, errors


@pytest.mark.parametrize("k,where,run", [(28, "end", 10_500), (28, "end", 7_000), (20, "end", 10_500), (28, "middle", 9_000), (12, "middle", 12_000)])
def test_blocks_that_give_up_and_are_repaired(ctx, k, where, run):
    """A SHORT run of one long code (here: 0xff under a unary-like code of up to k + 1 bits):
    the blocks it covers hit the first sweep's trip cap, the repair sweep settles them and the
    verification passes.  The speculative write and the host must judge that state by the same
    rule -- a long soak found the kernel declining ("blocks gave up") while the host kept its
    output ("verification passed"): all zeros, no error."""
    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(608)
    text = np.minimum(rng.geometric(0.5, size=2_500_000) - 1, k).astype(np.uint8)
    good = bytearray(O.encode(text)[4:])
    _, _, off = E.parse_header(bytes(good))
    a = len(good) - run if where == "end" else off + (len(good) - off) // 2 + 1234
    good[a : a + run] = b"\xff" * run
    want = O.decode(bytes(good))
    assert len(want) > 2_000_000
    for _ in range(2):
        assert ctx.decode(bytes(good)) == want


def test_cut_code_across_the_last_subsequence_boundary(ctx):
    """Truncated streams whose last 256-bit subsequence holds only 8 to 24 bits: when the
    code cut by the stream's end BEGINS in the subsequence before, the lane that runs off the
    stream must hand "the stream is over" to the last lane -- it once handed over bit 0, and the
    last lane decoded the cut code's tail as one more symbol (found by tests/soak/soak_fuzz.py).
    A Zipf-like source (many 9..13-bit codes, so that cut codes are common); every such
    truncation of a 3-block stream, checked against the oracle."""
    import entreepy_amd as E

    O = _oracle()
    rng = np.random.default_rng(2312)
    p = 1.0 / (1.0 + np.arange(255))
    text = (rng.choice(255, size=30_000, p=p / p.sum()) + 1).astype(np.uint8)
    et = O.encode(text)[4:]
    _, _, off = E.parse_header(et)
    base = off - (off & 3)  # subsequences are counted from the aligned word the body starts in
    checked = 0
    for tail in (1, 2, 3):
        for T in range(base + 64 + tail, len(et), 32):
            data = et[:T]
            assert ctx.decode(data) == O.decode(data), f"truncated to {T} bytes ({tail} in the last subsequence)"
            checked += 1
    assert checked > 1500


def test_device_built_decode_tables_equal_the_host_builders(ctx):
    """Every decode call fills its lookup tables on the device (k_build_dec_tables) from the
    host's plan; et_selftest_decode_tables builds them with the host-side reference builders as
    well (the ones tests/test_sanitizers.py checks against a brute-force decoder) and compares
    entry for entry.  Code tables of every shape: few / many symbols, flat, skewed, geometric
    (long codes, second-level tables, more long prefixes than table slots)."""
    import ctypes

    import entreepy_amd as E
    from entreepy_amd import _native as N

    rng = np.random.default_rng(4711)
    checked = 0
    for it in range(400):
        hist = np.zeros(256, dtype=np.uint64)
        k, mode = int(rng.integers(1, 257)), it % 5
        idx = rng.choice(256, size=k, replace=False)
        if mode == 0:
            hist[idx] = rng.integers(1, 4, size=k)
        elif mode == 1:
            hist[idx] = rng.integers(1, 100_000, size=k)
        elif mode == 2:
            hist[idx] = np.uint64(1) << rng.integers(0, 31, size=k).astype(np.uint64)  # up to 32-bit codes
        elif mode == 3:
            hist[idx] = 7
        else:
            hist[idx] = (1.6 ** np.minimum(np.arange(k), 40)).astype(np.uint64) + 1
        cb = E.Codebook.from_histogram(hist)
        if cb.raw.max_length > 32 or cb.raw.n_coded == 0:
            continue
        where = ctypes.c_int(0)
        rc = N.lib().et_selftest_decode_tables(ctx._h, ctypes.byref(cb.raw), ctypes.byref(where))
        assert rc == N.ET_OK, f"iteration {it} (mode {mode}, {k} symbols, longest {cb.raw.max_length}): tables differ in part {where.value}"
        checked += 1
    assert checked > 300


def test_phase_timings_are_consistent(ctx):
    """et_last_timings_of after device calls with timing on: the large kernels carry their own
    begin/end events (hipExtLaunchKernelGGL), the phases between them are differences of those.
    Encode: hist + scan + body = total; decode: sync + body = total, the first sweep is part of
    sync -- on the text path, and on the exhaustive path (no first sweep: marker events)."""
    import torch

    import entreepy_amd as E

    c = E.Context(0)
    c.use_torch_stream()
    c.enable_timing(True)
    for data, exhaustive in ((corpus.text_like(6_000_000, 3), False), (corpus.uniform(3_000_000, 4, 1, 256), True)):  # (255 symbols: the row walk; 200 would settle under the tree walk)
        text = torch.from_numpy(data).cuda()
        enc = torch.zeros(E.encode_bound(data.size) + 64, dtype=torch.uint8, device="cuda")
        dec = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            m = c.encode_device(text, enc)
            k = c.decode_device(enc[4:m], dec)
            te, td = c.timings("encode"), c.timings("decode")
            assert k == data.size and torch.equal(dec[:k], text)
            assert te["hist_ms"] > 0 and te["body_ms"] > 0 and te["scan_ms"] > 0
            assert abs(te["hist_ms"] + te["scan_ms"] + te["body_ms"] - te["total_ms"]) < 0.02 * te["total_ms"] + 1e-3
            assert td["exhaustive_sync"] == exhaustive
            assert td["body_ms"] > 0 and td["sync_ms"] > 0 and 0.0 < td["total_ms"] < 50.0
            assert abs(td["sync_ms"] + td["body_ms"] - td["total_ms"]) < 0.02 * td["total_ms"] + 1e-3
            if not exhaustive:
                assert 0 < td["sync_first_ms"] <= td["sync_ms"] * 1.001
    c.close()


def test_timing_of_the_write_kernel_alone(ctx):
    """et_ctx_enable_timing(ET_TIMING_DECODE_BODY): only the decode's write kernel carries its pair of events (what bench.py's
    timed regions use: a dispatch with events costs queue time on either side of it).  The decode's timings then hold
    body_ms, the host's time and the path flags, everything else 0; results are what they are with timing on or off; and
    switching back gives the full set again."""
    import torch

    import entreepy_amd as E

    c = E.Context(0)
    c.use_torch_stream()
    data = corpus.text_like(5_000_000, 21)
    text = torch.from_numpy(data).cuda()
    enc = torch.zeros(E.encode_bound(data.size) + 64, dtype=torch.uint8, device="cuda")
    dec = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
    c.enable_timing(True)
    m = c.encode_device(text, enc)
    k = c.decode_device(enc[4:m], dec)
    full = c.timings("decode")
    image = enc[:m].clone()
    c.enable_timing(c.TIMING_DECODE_BODY)
    for _ in range(3):
        dec.zero_()
        m2 = c.encode_device(text, enc)
        k = c.decode_device(enc[4:m2], dec)
        td = c.timings("decode")
        assert m2 == m and torch.equal(enc[:m], image) and k == data.size and torch.equal(dec[:k], text)
        assert 0.0 < td["body_ms"] < 5 * full["body_ms"] + 1.0 and td["tree_walk_sync"] and td["chained_write"]
        assert td["sync_ms"] == 0.0 and td["sync_first_ms"] == 0.0 and td["total_ms"] == 0.0 and td["scan_ms"] == 0.0
    c.enable_timing(True)
    m3 = c.encode_device(text, enc)
    k = c.decode_device(enc[4:m3], dec)
    te, td = c.timings("encode"), c.timings("decode")
    assert te["hist_ms"] > 0 and td["sync_ms"] > 0 and td["body_ms"] > 0 and abs(td["sync_ms"] + td["body_ms"] - td["total_ms"]) < 0.02 * td["total_ms"] + 1e-3
    c.close()


def test_header_and_zero_bit_tiles_share_the_seam_word(ctx):
    """All 256 byte values occur, so the reference drops the most frequent one (quirk Q1: code
    length 0) -- and the text STARTS with 6 MiB of it: more than 1024 tiles (a whole scan group and
    more) contribute no bits and all begin in the word that holds the header/body seam.  The scan
    kernel copies the header over that word; only one thread may zero it."""
    O = _oracle()
    head = np.full(6 << 20, 7, dtype=np.uint8)
    rest = np.tile(np.arange(256, dtype=np.uint8), 3000)
    data = np.concatenate([head, rest])
    ctx.set_tile_rounds(1)  # 4 KiB tiles: 1536 of them inside the run
    try:
        for _ in range(3):
            assert ctx.encode(data) == O.encode(data)
    finally:
        ctx.set_tile_rounds(0)


def test_declared_length_shorter_or_longer_than_the_body(ctx):
    """The header's length field (decode.zig:36-42) decides how many symbols come out: shorter
    than what the body holds -> exactly that many (also 0, 1, block and subsequence boundaries);
    longer -> everything the body holds, pad bits included if they happen to form a code (the format's
    own ambiguity).  Against the oracle's intended decoder."""
    O = _oracle()
    rng = np.random.default_rng(99)
    text = corpus.text_like(900_000, 17)
    et = bytearray(O.encode(text)[4:])
    lengths = [0, 1, 2, 255, 256, 257, 13_999, 14_000, 14_001, 450_000, 899_999, 900_000, 900_001, 5_000_000] + [int(x) for x in rng.integers(0, 900_000, size=20)]
    for n_decl in lengths:
        et[1:5] = int(n_decl).to_bytes(4, "big")
        want = O.decode(bytes(et), cap=max(n_decl, 900_000) + 64)
        assert min(n_decl, 900_000) <= len(want) <= min(n_decl, 900_002)  # (pad bits may hold one more short code)
        assert ctx.decode(bytes(et)) == want, f"declared {n_decl}"


def test_tree_walk_sync_is_what_a_decode_runs(ctx):
    """The synchronisation sweeps of an ordinary decode are the tree walk (et_treewalk.hip): timings say so, the
    table the device builds equals the host fill, and streams built to stress it -- every code length from 1 to
    24, a known first bit at every offset of a word, stream ends at every byte of a 512-bit lane -- decode to
    what the oracle decodes."""
    import ctypes

    import torch

    import entreepy_amd as E
    from entreepy_amd import _native as N

    O = _oracle()
    ctx.enable_timing(True)
    try:
        data = corpus.text_like(3_000_000, 99)
        et = O.encode(data)
        comp = torch.frombuffer(bytearray(et[4:]), dtype=torch.uint8).cuda()
        out = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
        assert ctx.decode_device(comp, out) == data.size
        t = ctx.timings("decode")
        assert t["tree_walk_sync"] and t["chained_write"] and not t["exhaustive_sync"]
        assert out[: data.size].cpu().numpy().tobytes() == data.tobytes()
    finally:
        ctx.enable_timing(False)
    # device-built table == host fill, for code tables of many shapes
    rng = np.random.default_rng(12)
    for trial in range(25):
        k = int(rng.integers(2, 257))
        h = np.zeros(256, dtype=np.uint64)
        h[rng.choice(256, size=k, replace=False)] = rng.integers(1, 1 << int(rng.integers(2, 28)), size=k)
        cb = E.Codebook.from_histogram(h)
        if cb.raw.max_length > 32:
            continue
        diff = ctypes.c_uint32(0)
        assert N.lib().et_selftest_treewalk_table(ctx._h, ctypes.byref(cb.raw), ctypes.byref(diff)) == N.ET_OK, diff.value
    # geometric counts: code lengths 1 .. 24; lengths of the stream that end it at every byte of a lane
    sym = np.repeat(np.arange(25, dtype=np.uint8) + 60, np.maximum(1, (1 << 24) >> np.arange(25)))
    rng.shuffle(sym)
    for cut in list(range(0, 70)) + [8191, 8192, 8193]:
        piece = sym[: sym.size - cut * 3]
        et = O.encode(piece)
        assert ctx.decode(et[4:]) == piece.tobytes(), cut
    # the body's first bit at every offset of a 4-byte word: slices of a device buffer at different alignments
    et = O.encode(corpus.text_like(400_000, 100))
    want = O.decode(et[4:])
    for shift in range(8):
        buf = torch.zeros(len(et) + 16, dtype=torch.uint8, device="cuda")
        buf[shift : shift + len(et) - 4] = torch.frombuffer(bytearray(et[4:]), dtype=torch.uint8).cuda()
        out = torch.empty(400_064, dtype=torch.uint8, device="cuda")
        m = ctx.decode_device(buf[shift : shift + len(et) - 4], out)
        assert out[:m].cpu().numpy().tobytes() == want, shift


def test_chained_write_windows_and_long_codes(ctx):
    """The write pass over chained lookup tables (k_dec_write_wave) where its special paths are: blocks that hold
    more symbols than the LDS stage (1- and 2-bit codes: up to four windows per block), a long-tailed alphabet whose
    rare symbols take two to four chained lookups, the same with the rare symbols made frequent in the STREAM (every
    lane meets several, also as its last codeword), and streams cut inside the last block."""
    import torch

    O = _oracle()
    rng = np.random.default_rng(77)
    cases = []
    cases.append(np.where(rng.random(3_000_000) < 0.9, 65, 66).astype(np.uint8))                      # 1-bit codes
    cases.append(rng.choice(np.array([1, 2, 3, 4], dtype=np.uint8), size=2_000_000, p=[0.4, 0.3, 0.2, 0.1]))  # 1..3 bits
    cases.append(corpus.enwik_like(4_000_000, 5))
    # a geometric alphabet (code lengths 1 .. 24) in which the tail is as frequent as the head
    sym = np.repeat(np.arange(25, dtype=np.uint8) + 60, np.maximum(1, (1 << 24) >> np.arange(25)))
    rng.shuffle(sym)
    cases.append(sym[:3_000_000])
    ctx.enable_timing(True)
    try:
        for data in cases:
            et = O.encode(data)
            comp = torch.frombuffer(bytearray(et[4:]), dtype=torch.uint8).cuda()
            out = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
            assert ctx.decode_device(comp, out) == data.size
            t = ctx.timings("decode")
            assert t["chained_write"] or t["fixed_sync"]  # (two 1-bit codewords go by arithmetic since round 4: the windows of such a block, tests/test_gpu_fixedsync.py's ET_NO_FIXED_SYNC child)
            assert out[: data.size].cpu().numpy().tobytes() == data.tobytes()
            for cut in (1, 2, 3, 5, 64, 257, 8191, 8200):  # truncated bodies: whatever the oracle makes of them
                short = et[4 : len(et) - cut]
                assert ctx.decode(short) == O.decode(short), cut
    finally:
        ctx.enable_timing(False)
    # the tail symbols FREQUENT in the stream: encode with the skewed table, decode a body assembled by hand
    import entreepy_amd as E

    hist = np.zeros(256, dtype=np.uint64)
    hist[60:85] = np.maximum(1, (1 << 24) >> np.arange(25)).astype(np.uint64)
    cb = E.Codebook.from_histogram(hist)
    text = (rng.integers(0, 25, size=400_000) + 60).astype(np.uint8)  # uniform over the 25 symbols: mean length ~13 bits
    lens = cb.length[text].astype(np.int64)
    ends = np.cumsum(lens)
    total = int(ends[-1])
    bits = np.zeros(total + 64, dtype=np.uint8)
    starts = ends - lens
    for k in range(int(lens.max())):  # bit k of every code that has one
        has = lens > k
        bits[starts[has] + k] = (cb.data[text[has]] >> (lens[has] - 1 - k).astype(np.uint32)) & 1
    body = np.packbits(bits[: (total + 7) // 8 * 8])
    d_body = torch.from_numpy(body).cuda()
    d_out = torch.empty(text.size + 64, dtype=torch.uint8, device="cuda")
    assert ctx.decode_body_device(cb, d_body, text.size, d_out) == text.size
    assert d_out[: text.size].cpu().numpy().tobytes() == text.tobytes()


def test_scan_epochs_run_out_and_start_again(ctx):
    """k_scan_fused tells this launch's published group totals from older ones by a 16-bit epoch; after 65535 launches on
    a context the words are zeroed and the epochs start again.  Many more than that many scans (an encode has one, a
    decode has one) on one context, all with more than one group in the decode's scan (> 1024 blocks), must keep giving
    the right bytes -- also across a change of size, which reallocates the words."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    c = E.Context(0)
    c.use_torch_stream()
    small = corpus.text_like(30_000, 7)
    big = corpus.text_like(24_000_000, 8)  # ~14 MB packed: ~1700 blocks, two scan groups; 366 encode tiles
    want_small, want_big = O.encode(small), O.encode(big)
    d_small, d_big = torch.from_numpy(small).cuda(), torch.from_numpy(big).cuda()
    enc = torch.zeros(E.encode_bound(big.size) + 64, dtype=torch.uint8, device="cuda")
    dec = torch.empty(big.size + 64, dtype=torch.uint8, device="cuda")

    def round_trip(d_text, want, check):
        m = c.encode_device(d_text, enc)
        k = c.decode_device(enc[4:m], dec)
        assert m == len(want) and k == d_text.numel()
        if check:
            assert enc[:m].cpu().numpy().tobytes() == want
            assert torch.equal(dec[:k], d_text)

    round_trip(d_big, want_big, True)
    for i in range(33_500):  # 67 000 scans
        round_trip(d_small, want_small, i % 2000 == 0)
        if i % 4000 == 1:
            round_trip(d_big, want_big, True)
    round_trip(d_big, want_big, True)
    c.close()


def test_histogram_host_and_device_copies_agree(ctx):
    """et_histogram_host (the reduction's stores into pinned memory, polled) and et_histogram_device's device copy are
    the same counts, numpy's; an empty text gives zeros on both sides."""
    import torch

    for n in (1, 4095, 1 << 20, 5_000_003):
        data = corpus.text_like(n, n)
        d = torch.from_numpy(data).cuda()
        hist = torch.zeros(256, dtype=torch.int64, device="cuda")
        ctx.histogram_device(d, hist)
        want = np.bincount(data, minlength=256).astype(np.uint64)
        assert np.array_equal(ctx.histogram_host(), want)
        assert np.array_equal(hist.cpu().numpy().astype(np.uint64), want)
    hist = torch.ones(256, dtype=torch.int64, device="cuda")
    ctx.histogram_device(torch.empty(0, dtype=torch.uint8, device="cuda"), hist)
    assert int(hist.sum()) == 0 and int(ctx.histogram_host().sum()) == 0


def test_polled_hand_overs_fall_back_to_a_stream_wait(ctx):
    """The calls poll pinned words for the histogram, the header and the decode's report; when the stream is held up by
    somebody else's work for longer than their patience (0.1 - 0.2 s) they fall back to a stream wait and still
    deliver: half a second of matrix products is put in front of an encode and of a decode on the same stream."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    c = E.Context(0)
    c.use_torch_stream()
    data = corpus.text_like(3_000_000, 61)
    want = O.encode(data)
    text = torch.from_numpy(data).cuda()
    enc = torch.zeros(E.encode_bound(data.size) + 64, dtype=torch.uint8, device="cuda")
    dec = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
    a = torch.randn(8192, 8192, device="cuda")

    def hold_the_stream():
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        x = a
        for _ in range(120):
            x = (x @ a) * 1e-2
        t1.record()
        return t0, t1

    t0, t1 = hold_the_stream()
    m = c.encode_device(text, enc)  # its histogram arrives long after the poll's patience
    assert t0.elapsed_time(t1) > 150.0, "the stream was not held long enough to exercise the fallback"
    assert enc[:m].cpu().numpy().tobytes() == want
    t0, t1 = hold_the_stream()
    k = c.decode_device(enc[4:m], dec)
    assert t0.elapsed_time(t1) > 150.0
    assert k == data.size and torch.equal(dec[:k], text)
    c.close()


def test_two_contexts_on_two_threads_device_calls(ctx):
    """Contexts are independent: two host threads, each with its own context and stream, encode and decode different
    texts at the same time; every result equals the oracle's."""
    import threading

    import torch

    import entreepy_amd as E

    O = _oracle()
    texts = [corpus.text_like(2_500_000, 71), corpus.enwik_like(2_000_000, 72)]
    wants = [O.encode(t) for t in texts]
    errors = []

    def worker(i):
        try:
            c = E.Context(0)
            d = torch.from_numpy(texts[i]).cuda()
            enc = torch.zeros(E.encode_bound(texts[i].size) + 64, dtype=torch.uint8, device="cuda")
            dec = torch.empty(texts[i].size + 64, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for _ in range(40):
                m = c.encode_device(d, enc)
                k = c.decode_device(enc[4:m], dec)
                assert m == len(wants[i]) and k == texts[i].size
            torch.cuda.synchronize()
            assert enc[:m].cpu().numpy().tobytes() == wants[i]
            assert torch.equal(dec[:k], d)
            c.close()
        except Exception as e:  # noqa: BLE001 (reported in the main thread)
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errors, errors
    assert not any(t.is_alive() for t in ts)


def test_hand_made_tables_with_codes_nobody_has(ctx):
    """Dictionaries that leave bit patterns without a symbol (no encoder makes them): a valid stream decodes to its
    text on every path -- a skewed table (tree walk + chained tables, the patterns become leaves), a flat one (the
    exhaustive path, which keeps the older write kernel for such trees) -- and a stream that DOES contain the
    patterns gives bounded output and no fault."""
    import torch

    import entreepy_amd as E

    rng = np.random.default_rng(123)

    def pack(data_t, len_t, text):
        lens = len_t[text].astype(np.int64)
        ends = np.cumsum(lens)
        total = int(ends[-1])
        bits = np.zeros(total + 64, dtype=np.uint8)
        starts = ends - lens
        for k in range(int(lens.max())):
            has = lens > k
            bits[starts[has] + k] = (data_t[text[has]] >> (lens[has] - 1 - k).astype(np.uint32)) & 1
        return np.packbits(bits[: (total + 7) // 8 * 8])

    # flat: 200 symbols with 8-bit codes 0..199; skewed: codes 0, 10, 110, 1110xxxx (16 of them), pattern 1111.... unused
    flat_d, flat_l = np.zeros(256, np.uint32), np.zeros(256, np.uint8)
    flat_d[:200], flat_l[:200] = np.arange(200), 8
    skew_d, skew_l = np.zeros(256, np.uint32), np.zeros(256, np.uint8)
    skew_d[65], skew_l[65] = 0b0, 1
    skew_d[66], skew_l[66] = 0b10, 2
    skew_d[67], skew_l[67] = 0b110, 3
    for i in range(16):
        skew_d[100 + i], skew_l[100 + i] = (0b1110 << 4) | i, 8
    for data_t, len_t, syms in ((flat_d, flat_l, np.arange(200)), (skew_d, skew_l, np.array([65, 66, 67] + list(range(100, 116))))):
        cb = E.Codebook.from_tables(data_t, len_t)
        text = syms[rng.integers(0, syms.size, size=600_000)].astype(np.uint8)
        body = pack(data_t, len_t, text)
        d_body = torch.from_numpy(body).cuda()
        out = torch.empty(text.size + 64, dtype=torch.uint8, device="cuda")
        assert ctx.decode_body_device(cb, d_body, text.size, out) == text.size
        assert out[: text.size].cpu().numpy().tobytes() == text.tobytes()
        # junk: every byte value, so also the patterns no symbol has
        junk = torch.from_numpy(rng.integers(0, 256, size=body.size, dtype=np.uint8)).cuda()
        m = ctx.decode_body_device(cb, junk, text.size, out)
        torch.cuda.synchronize()
        assert 0 <= m <= text.size


def _sparse_dictionary():
    """A prefix-free table no encoder makes: two 2-bit codes and a hundred 16-bit codes 1iiiiiii00000000 -- completed with a
    leaf for every bit pattern nobody has, its tree has ~900 internal nodes, beyond the tree walk's table (255)."""
    data_t, len_t = np.zeros(256, np.uint32), np.zeros(256, np.uint8)
    data_t[32], len_t[32] = 0b00, 2
    data_t[101], len_t[101] = 0b01, 2
    for i in range(100):
        data_t[120 + i], len_t[120 + i] = 0x8000 | (i << 8), 16
    return data_t, len_t, np.array([32, 101] + list(range(120, 220)))


@pytest.mark.parametrize("n", [4_000, 60_000, 2_000_000])
def test_fallback_sweeps_and_write_are_what_a_sparse_dictionary_runs(ctx, n):
    """The round-1 kernels are still the decoder of dictionaries outside the tree walk's domain: a whole-stream decode with
    such a table reports tree_walk_sync = chained_write = False (at the three sizes that pick k_dec_sync / k_dec_write alone,
    k_dec_sync_reg + k_dec_write_reg, and k_dec_sync_reg2), and returns the text the oracle packed."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    data_t, len_t, syms = _sparse_dictionary()
    with pytest.raises(E.EntreepyError):  # (not a tree the walk can hold)
        import ctypes

        from entreepy_amd import _native as N

        k = ctypes.c_uint32(0)
        E.codec._check(N.lib().et_treewalk_table(ctypes.byref(E.Codebook.from_tables(data_t, len_t).raw), None, 0, ctypes.byref(k)))
    rng = np.random.default_rng(n)
    cb = E.Codebook.from_tables(data_t, len_t)
    text = syms[np.minimum(rng.integers(0, 300, size=n), syms.size - 1) % syms.size].astype(np.uint8)
    text[rng.random(n) < 0.5] = 32  # half of it the 2-bit code: the stream re-synchronises
    body, end = O.pack_body(data_t, len_t, text)
    d_body = torch.from_numpy(np.frombuffer(body, dtype=np.uint8).copy()).cuda()
    out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    ctx.enable_timing(True)
    try:
        assert ctx.decode_body_device(cb, d_body, n, out) == n
        t = ctx.timings("decode")
    finally:
        ctx.enable_timing(False)
    torch.cuda.synchronize()
    assert not t["tree_walk_sync"] and not t["chained_write"] and not t["exhaustive_sync"]
    assert out[:n].cpu().numpy().tobytes() == text.tobytes()


def test_fallback_range_sync_is_what_a_sparse_dictionary_runs(ctx):
    """et_decode_range_sync / _write with the same dictionary: ranges of a split stream take the round-1 sweeps (info says so),
    a wrong first guess is repaired, and the pieces concatenate to the text."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    data_t, len_t, syms = _sparse_dictionary()
    cb = E.Codebook.from_tables(data_t, len_t)
    rng = np.random.default_rng(9)
    n = 400_000
    text = syms[rng.integers(0, syms.size, size=n)].astype(np.uint8)
    text[rng.random(n) < 0.5] = 101
    body, end = O.pack_body(data_t, len_t, text)
    stream = torch.from_numpy(np.frombuffer(body, dtype=np.uint8).copy()).cuda()
    n_blocks = (stream.numel() + 8191) // 8192
    cuts = [0, (n_blocks // 3) * 8192, (2 * n_blocks // 3) * 8192, stream.numel()]
    ctxs, infos = [], []
    for r in range(3):
        c = E.Context(0)
        c.use_torch_stream()
        infos.append(c.decode_range_sync(cb, stream, cuts[r], cuts[r + 1], 0 if r == 0 else -1))
        assert not infos[-1]["tree_walk"]
        ctxs.append(c)
    for _ in range(5):
        prev, wrong = 0, []
        for i, inf in enumerate(infos):
            if inf["start_bit"] != prev:
                wrong.append((i, prev))
            prev = inf["exit_bit"]
        if not wrong:
            break
        for i, w in wrong:
            infos[i] = ctxs[i].decode_range_sync(cb, stream, cuts[i], cuts[i + 1], w)
    else:
        raise AssertionError("did not settle")
    pieces = []
    for c, inf in zip(ctxs, infos):
        buf = torch.empty(inf["n_symbols"] + 64, dtype=torch.uint8, device="cuda")
        m = c.decode_range_write(inf["n_symbols"], buf)
        torch.cuda.synchronize()
        pieces.append(buf[:m].cpu().numpy())
        c.close()
    assert np.concatenate(pieces)[:n].tobytes() == text.tobytes()


def test_write_by_quarters_windows_ragged_ends_and_clamps(ctx):
    """k_dec_write_wave: a wavefront owns a quarter of an 8 KiB block (64 subsequences) and a 4 KiB stage.  Streams whose
    quarters decode to MORE than a stage holds (a 1-bit code: up to 16 K symbols per quarter -> several windows per
    quarter), streams that end inside every quarter of their last block and inside a quarter's first / last subsequence,
    declared lengths that cut the output inside a quarter, at a quarter's edge and at a 16-byte chunk's edge -- all against
    the oracle's intended decoder; and output buffers at every 16-byte phase."""
    import torch

    O = _oracle()
    rng = np.random.default_rng(2024)
    # 1-bit code for one symbol, a tail of others: ~7 symbols per byte of stream
    for p_hot, n in ((0.995, 3_000_000), (0.9, 1_500_000), (0.6, 800_000)):
        base = np.where(rng.random(n) < p_hot, 120, corpus.text_like(n, int(p_hot * 1000))).astype(np.uint8)
        et = O.encode(base)
        body = len(et) - 4
        assert ctx.decode(et[4:]) == base.tobytes(), p_hot
        # the stream cut inside every quarter of a late block, and around subsequence edges (what is left decodes as the oracle says)
        for cut in (1, 31, 32, 33, 2047, 2048, 2049, 4096 + 5, 6144 - 1, 8192 - 32, 8192, 8192 + 1):
            if cut + 64 < body:
                piece = et[4 : len(et) - cut]
                assert ctx.decode(piece) == O.decode(piece), (p_hot, cut)
    # declared lengths inside / at the edge of quarters and 16-byte chunks
    text = corpus.text_like(700_000, 5)
    et = bytearray(O.encode(text))
    comp = torch.frombuffer(bytearray(et[4:]), dtype=torch.uint8).cuda()
    full = O.decode(bytes(et[4:]))
    for declared in (0, 1, 15, 16, 17, 3493, 3500, 4096, 4097, 13970, 13972 * 3 + 7, len(full) - 1, len(full)):
        hdr = bytearray(et[4:])
        hdr[1:5] = int(declared).to_bytes(4, "big")
        want = O.decode(bytes(hdr))
        assert ctx.decode(bytes(hdr)) == want, declared
    # the output buffer at every phase of a 16-byte chunk is refused or right: the C ABI asks for 16-byte alignment
    out = torch.empty(len(full) + 64, dtype=torch.uint8, device="cuda")
    m = ctx.decode_device(comp, out)
    assert m == len(full) and out[:m].cpu().numpy().tobytes() == full


def test_device_calls_are_ordered_on_torchs_current_stream():
    """A fresh Context, no use_torch_stream(), no synchronize anywhere: the input is still being PRODUCED on torch's stream (a
    chain of large element-wise kernels) when encode_device / decode_device are called, on the default stream and inside a
    torch.cuda.stream(...) block -- the calls run on whatever stream is current, so they see finished tensors and the decode
    sees the finished image (the soak of round 3 raced here: a context used to sit on a private stream unless told otherwise)."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    host = corpus.text_like(24 << 20, 77)
    want = O.encode(host.tobytes())
    side = torch.cuda.Stream()
    for stream in (None, side):
        c = E.Context(0)  # fresh: default behaviour
        try:
            base = torch.from_numpy(host).cuda()
            enc = torch.zeros(E.encode_bound(host.size) + 64, dtype=torch.uint8, device="cuda")
            dec = torch.zeros(host.size + 64, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
                text = base ^ 0x5A  # a few ms of kernels on the current stream; the last one restores the text
                for _ in range(40):
                    text = (text ^ 0xFF) ^ 0xFF
                text = text ^ 0x5A
                n = c.encode_device(text, enc)  # no synchronize in front of it
                m = c.decode_device(enc[4:n], dec)  # nor here
                got = enc[:n].clone()
            torch.cuda.synchronize()
            assert got.cpu().numpy().tobytes() == want
            assert m == host.size and dec[:m].cpu().numpy().tobytes() == host.tobytes()
        finally:
            c.close()

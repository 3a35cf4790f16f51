"""GPU: streams with more than 128 symbols per 256-bit subsequence -- a dominant symbol with a 1- or 2-bit codeword -- take the write
pass's instantiation that walks a quarter once, every lane into a strip of its own that leaves for the output every two stream words
(k_dec_write_wave<8, true>, walk_write_chain<3>: csrc/et_kernels.hip), instead of once per 4 KiB window of the quarter's output.
Against the oracle and the text, through the C ABI; ET_NO_STRIPS=1 keeps the windows pinned."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT
from tests.test_gpu_rowsync import _oracle, _timed_decode

pytestmark = pytest.mark.gpu


def sparse(n, p_zero, seed, others=254):
    """n bytes: a share p_zero of zeros, the rest uniform over `others` other values."""
    rng = np.random.default_rng(seed)
    data = rng.integers(1, 1 + others, size=n).astype(np.uint8)
    data[rng.random(n) < p_zero] = 0
    return data


@pytest.mark.parametrize("p_zero", [0.92, 0.95, 0.97, 0.995])
def test_sparse_streams_walk_once_into_strips(ctx, p_zero):
    O = _oracle()
    for n in (70_001, 1_000_003):
        data = sparse(n, p_zero, int(p_zero * 1000) + n % 7)
        et = O.encode(data.tobytes())
        assert ctx.encode(data.tobytes()) == et
        back, t = _timed_decode(ctx, et)
        assert back == data.tobytes(), (p_zero, n)
        assert t["strips_write"] and t["tree_walk_sync"] and t["chained_write"], t


def test_few_symbols_with_short_codes(ctx):
    """Three and four symbols with codewords of 1..3 bits (up to 256 symbols per subsequence: every two-word strip is full),
    and 40 symbols of which one takes 95 %: sub-tables behind the root table inside strips."""
    O = _oracle()
    rng = np.random.default_rng(3)
    cases = [rng.choice(np.array([7, 8, 9], dtype=np.uint8), size=900_001, p=[0.8, 0.15, 0.05]),
             rng.choice(np.array([1, 2, 3, 4], dtype=np.uint8), size=1_200_000, p=[0.7, 0.2, 0.07, 0.03])]
    tail = np.repeat(np.arange(40, dtype=np.uint8) + 100, np.maximum(1, (1 << 16) >> np.arange(40)))
    long_tail = np.concatenate([np.full(2_000_000, 100, dtype=np.uint8), tail])
    rng.shuffle(long_tail)
    cases.append(long_tail)
    for data in cases:
        et = O.encode(data.tobytes())
        back, t = _timed_decode(ctx, et)
        assert back == data.tobytes() and t["strips_write"], t


def test_sizes_truncations_and_declared_lengths(ctx):
    """Streams that end everywhere around a subsequence and a block; truncated images against the oracle (the last quarters are edge
    quarters: windows, inside the same kernel); declared counts shorter than the body (clamped quarters: windows too)."""
    import torch

    import entreepy_amd as E

    O = _oracle()
    base = sparse(400_000, 0.93, 11)
    for n in (1, 2, 150, 151, 200, 9_000, 46_000, 46_100, 46_200, 100_000, 399_999):
        et = O.encode(base[:n].tobytes())
        assert ctx.decode(et[4:]) == O.decode(et[4:]), n  # (= the text but for a lone symbol value, which encodes to the bare header: Q2)
        if len(set(base[:n].tolist())) > 1:
            assert O.decode(et[4:]) == base[:n].tobytes(), n
    et = O.encode(base.tobytes())[4:]
    _, _, off = E.parse_header(et)
    for cut in list(range(1, 40)) + [8190, 8193, 20_000]:
        assert ctx.decode(et[:-cut]) == O.decode(et[:-cut]), cut
    cb = E.Codebook.from_histogram(np.bincount(base, minlength=256).astype(np.uint64))
    for start_bit in (0, 5):
        body, end_bit = O.pack_body(cb.data, cb.length, base, start_bit)
        for shift in (0, 3):
            buf = torch.zeros(len(body) + 64, dtype=torch.uint8, device="cuda")
            buf[16 + shift : 16 + shift + len(body)] = torch.frombuffer(bytearray(body), dtype=torch.uint8).cuda()
            out = torch.full((base.size + 256,), 0xEE, dtype=torch.uint8, device="cuda")
            for n_decl in (base.size, base.size - 1, 300_000, 65_537, 17, 1):
                out.fill_(0xEE)
                m = ctx.decode_body_device(cb, buf[16 + shift : 16 + shift + (end_bit + 7) // 8], n_decl, out, start_bit)
                got = out.cpu().numpy()
                assert m == n_decl and got[:m].tobytes() == base[:m].tobytes(), (start_bit, shift, n_decl)
                assert (got[m + 16 :] == 0xEE).all(), (start_bit, shift, n_decl)


def test_dense_and_ordinary_quarters_in_one_stream(ctx):
    """A stream whose header promises many symbols per subsequence but whose middle is ordinary text-like bytes: quarters that fit the
    stage stage as ever, the others go into strips, in one launch."""
    import torch

    import entreepy_amd as E

    rng = np.random.default_rng(5)
    mid = rng.integers(1, 255, size=60_000).astype(np.uint8)
    data = np.concatenate([np.zeros(3_000_000, dtype=np.uint8), mid, np.zeros(2_000_000, dtype=np.uint8), mid[:777], np.zeros(10, dtype=np.uint8)])
    text = torch.from_numpy(data).cuda()
    enc = torch.zeros(E.encode_bound(data.size) + 64, dtype=torch.uint8, device="cuda")
    dec = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
    ctx.use_torch_stream()
    ln = ctx.encode_device(text, enc)
    ctx.enable_timing(True)
    try:
        m = ctx.decode_device(enc[4:ln], dec)
        t = ctx.timings("decode")
    finally:
        ctx.enable_timing(False)
    assert t["strips_write"] and m == data.size and torch.equal(dec[: data.size], text)


def test_large_sparse_stream(ctx):
    """256 MiB, 97 % zeros: 20 000 quarters of ~13 000 symbols each, compared on the device."""
    import torch

    import entreepy_amd as E

    n = 256 << 20
    g = torch.Generator(device="cuda")
    g.manual_seed(97)
    text = torch.randint(1, 255, (n,), generator=g, device="cuda", dtype=torch.int16).to(torch.uint8)
    text[torch.rand(n, generator=g, device="cuda") < 0.97] = 0
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
    assert t["strips_write"] and m == n and torch.equal(dec[:n], text)


def test_windows_still_decode_these_streams():
    """ET_NO_STRIPS=1 (a child process: the switch is read once): the same streams, a quarter walked once per window."""
    code = (
        "import numpy as np, entreepy_amd as E\n"
        "from tests.test_gpu_strips import sparse\n"
        "from oracle import oracle as O\n"
        "c = E.Context(0); c.enable_timing(True)\n"
        "for p in (0.9, 0.99):\n"
        "    d = sparse(600_001, p, 5)\n"
        "    et = O.encode(d.tobytes())\n"
        "    assert c.decode(et[4:]) == d.tobytes()\n"
        "    t = c.timings('decode')\n"
        "    assert t['chained_write'] and not t['strips_write'], t\n"
        "print('ok')\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, ET_NO_STRIPS="1"), timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]

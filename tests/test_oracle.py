"""CPU: the oracle against every vector we hold for the path (no GPU)."""
import hashlib
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import oracle as O
from tests.conftest import GOLDEN

SURVEY_TEST_TXT_HEX = "e7c0de01060000002f0a05e208148409e860bd44021140f2f812934243dc619f87efccfe8c37dd9fc700"
SURVEY_SHA = {
    "test.txt": "761b5bc3dcb9d8487eaa764b7c1b207774caff78a464b229c56f939417af764d",
    "nice.shakespeare.txt": "795f27fd81733435fbaa1e58d260950eec57f800652245464b7e3209407a2409",
    "a_midsummer_nights_dream.txt": "d152197f8c5ee87c68ca812929ebc92ffd74b0fecbdd6c491b203d19479eb97b",
}


def test_readme_known_answer(res_files):
    """README.md:51 -- the only exact number the reference publishes."""
    assert len(O.encode(res_files["nice.shakespeare.txt"])) == 374


def test_survey_vectors(res_files):
    assert O.encode(res_files["test.txt"]).hex() == SURVEY_TEST_TXT_HEX
    for name, sha in SURVEY_SHA.items():
        assert hashlib.sha256(O.encode(res_files[name])).hexdigest() == sha
    assert len(O.encode(res_files["a_midsummer_nights_dream.txt"])) == 66312


def test_hand_derived_codes(res_files):
    """D=00 _=01 A=10 E=110 B=1111 \\n=11100 C=11101, derived by hand from encode.zig:102-135."""
    data, length, order = O.build_dict(O.histogram(res_files["test.txt"]))
    want = {"D": "00", "_": "01", "A": "10", "E": "110", "B": "1111", "\n": "11100", "C": "11101"}
    for ch, bits in want.items():
        s = ord(ch)
        assert length[s] == len(bits) and format(int(data[s]), "b").zfill(len(bits)) == bits


@pytest.mark.parametrize("name", list(SURVEY_SHA))
def test_reference_round_trips_through_literal_decoder(res_files, name):
    """src/test.zig:35-72: encode -> decode(encoded[4..]) == input, using the LITERAL
    restatement of decode.zig (quirks and all) and the intended inverse."""
    et = O.encode(res_files[name])
    assert O.decode_ref(et[4:]) == res_files[name]
    assert O.decode(et[4:]) == res_files[name]


def test_committed_golden_files(res_files):
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        manifest = json.load(f)
    for name, m in manifest.items():
        if "input" in m:
            text = res_files[name]
            with open(os.path.join(GOLDEN, name + ".et"), "rb") as f:
                et = f.read()
        else:
            text, et = bytes.fromhex(m["input_hex"]), bytes.fromhex(m["et_hex"])
        assert O.encode(text) == et and hashlib.sha256(et).hexdigest() == m["sha256"], name


def test_edge_vectors():
    assert O.encode(b"aaaa").hex() == "e7c0de010000000004"  # header only
    e = O.encode(b"ab" * 10)
    assert len(e) == 17 and e.hex().startswith("e7c0de010100000014")
    with pytest.raises(O.OracleError) as ei:
        O.encode(b"")
    assert ei.value.status == O.QUEUE_EMPTY  # Q5


def test_quirk_q1_256_symbols_drop_the_most_frequent():
    hist = np.arange(1, 257, dtype=np.uint64)  # byte 255 is the most frequent
    data, length, order = O.build_dict(hist)
    assert length[255] == 0 and (length[:255] > 0).all() and len(order) == 255
    hist[:] = 7  # all tied: the highest byte value is last in (count, byte) order
    _, length, _ = O.build_dict(hist)
    assert length[255] == 0 and (length[:255] > 0).all()
    text = bytes(range(256)) * 2
    assert O.encode(text)[4] == 254  # D byte


def test_quirk_q3_u32_truncation_of_long_codes():
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    hist = np.zeros(256, dtype=np.uint64)
    hist[:40] = fib
    data, length, _ = O.build_dict(hist)
    assert length.max() == 39
    # the deepest leaves lost their leading path bits: data < 2**32 by construction,
    # and the emitted bit k is bit (k mod 32) of it
    text = bytes([0, 1, 39, 0])
    body, end = O.pack_body(data, length, text)
    bits = "".join(format(b, "08b") for b in body)[:end]
    want = ""
    for s in text:
        want += "".join(str((int(data[s]) >> ((j - 1) & 31)) & 1) for j in range(int(length[s]), 0, -1))
    assert bits == want


def test_quirk_q4_length_field_wraps():
    data, length, _ = O.build_dict(O.histogram(b"abc"))
    h = O.write_header(data, length, (1 << 34) + 5)
    assert h[5:9] == (5).to_bytes(4, "big")


def test_quirk_q6_q7_literal_decoder(res_files):
    with pytest.raises(O.OracleError) as ei:
        O.decode_ref(O.encode(b"a\x00b\x00\x00ccc")[4:])
    assert ei.value.status == O.HANG
    t = res_files["a_midsummer_nights_dream.txt"] + b"eee"
    et = O.encode(t)
    assert len(t) - len(O.decode_ref(et[4:])) == 3  # three tail symbols dropped
    assert O.decode(et[4:]) == t


def test_pack_body_matches_encode_and_is_linear(res_files):
    """encode == header || pack_body; packing shards at their bit offsets and OR-ing
    equals packing the whole (the property the multi-GPU concat relies on)."""
    text = res_files["a_midsummer_nights_dream.txt"]
    data, length, _ = O.build_dict(O.histogram(text))
    header = O.write_header(data, length, len(text))
    body, end = O.pack_body(data, length, text)
    assert header + body == O.encode(text)
    cut = 50001
    a, end_a = O.pack_body(data, length, text[:cut])
    b, end_b = O.pack_body(data, length, text[cut:], start_bit=end_a, cap=len(body) + 8)
    merged = np.zeros(len(body), dtype=np.uint8)
    merged[: len(a)] |= np.frombuffer(a, dtype=np.uint8)
    merged[: len(b)] |= np.frombuffer(b, dtype=np.uint8)[: len(body)]
    assert end_b == end and merged.tobytes() == body


@settings(max_examples=200, deadline=None)
@given(st.binary(min_size=1, max_size=2000))
def test_property_intended_decoder_inverts_encode(data):
    et = O.encode(data)
    out = O.decode(et[4:])
    if len(set(data)) == 1:
        assert out == b"" and len(et) == 9
    else:
        assert out == data


@settings(max_examples=100, deadline=None)
@given(st.lists(st.integers(1, 127), min_size=2, max_size=3000))
def test_property_literal_decoder_on_its_valid_domain(symbols):
    """No NUL, short codes: wherever the literal decoder returns all symbols it must
    agree with the input; otherwise it may only drop a tail (Q7)."""
    data = bytes(symbols)
    if len(set(data)) < 2:
        return
    et = O.encode(data)
    out = O.decode_ref(et[4:])
    assert data.startswith(out) and len(data) - len(out) < 32


def test_format_file_size():
    assert O.format_file_size(477) == "477 B"
    assert O.format_file_size(112541) == "109.90 KB"
    assert O.format_file_size(5458199) == "5.21 MB"


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_fast_cpu_variant_equals_the_restatement(threads):
    """oracle/et_cpu_fast.c (bench.py's all-cores CPU baseline): same .et bytes as the
    restatement, and its chunk-parallel decode (walk, mark, merge at the true starts)
    returns the input -- on text, a flat code (nothing to re-synchronise on: the merge
    must still find the true boundaries), all 256 values (Q1), tiny and one-symbol inputs."""
    from oracle import cpu_fast as F
    from tests import corpus

    cases = [corpus.text_like(1_200_000, 1), corpus.uniform(700_000, 2, 0, 256), corpus.uniform(600_000, 3, 1, 201),
             corpus.text_like(100, 4), np.full(5000, 65, np.uint8), corpus.uniform(300_000, 5, 7, 9)]
    for data in cases:
        want = O.encode(data)
        assert F.encode(data, threads) == want
        assert F.decode(want[4:], threads) == O.decode(want[4:])


def test_fast_cpu_variant_equals_the_restatement_at_64_mib():
    """The 1 GiB byte-exact GPU test (test_gpu_configs.py) compares with the fast CPU variant; this ties that variant to the
    restatement at a size where every code path of its chunking is long since in play (64 MiB of the bench's text stream, 8 and
    3 threads: chunks that do not divide the stream evenly), beside the small cases above."""
    from oracle import cpu_fast as F
    from tests import corpus

    data = corpus.text_like(64 << 20, 0x5EED0004)
    want = O.encode(data)
    for threads in (8, 3):
        assert F.encode(data, threads) == want
    assert F.decode(want[4:], 8) == data.tobytes()

"""GPU: the C++ `entreepy` CLI (main.zig surface) and the torch.distributed (RCCL) sharded path."""
import hashlib
import os
import socket
import subprocess

import numpy as np
import pytest

from tests import corpus
from tests.conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "entreepy_amd", "entreepy")


def test_cli_compress_decompress_round_trip(tmp_path, res_files):
    from oracle import oracle as O

    for name, text in res_files.items():
        src = tmp_path / name
        src.write_bytes(text)
        et = tmp_path / (name + ".et")
        r = subprocess.run([EXE, "c", str(src), "-o", str(et)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert et.read_bytes() == O.encode(text)
        with open(os.path.join(GOLDEN, name + ".et"), "rb") as f:
            assert et.read_bytes() == f.read()
        assert r.stderr.strip() == f"{O.format_file_size(len(text))} => {O.format_file_size(et.stat().st_size)}"  # encode.zig:334
        back = tmp_path / ("back_" + name)
        r = subprocess.run([EXE, "d", str(et), "-o", str(back)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert back.read_bytes() == text
        assert r.stderr.strip() == f"{O.format_file_size(et.stat().st_size - 4)} => {O.format_file_size(len(text))}"  # decode.zig:217


def test_cli_default_names_dry_run_print_and_debug(tmp_path, res_files):
    from oracle import oracle as O

    text = res_files["test.txt"]
    src = tmp_path / "t.txt"
    src.write_bytes(text)
    assert subprocess.run([EXE, "compress", str(src)], capture_output=True).returncode == 0  # any word starting with c (main.zig:123)
    assert (tmp_path / "t.txt.et").read_bytes() == O.encode(text)  # default [file].et (main.zig:156-158)
    assert subprocess.run([EXE, "d", str(tmp_path / "t.txt.et")], capture_output=True).returncode == 0
    assert (tmp_path / "decoded_t.txt").read_bytes() == text  # decoded_[file minus .et] (main.zig:159-169)
    # -t: nothing written; decode reports "=> 0 B" (decode.zig:185-188)
    out = tmp_path / "never.et"
    r = subprocess.run([EXE, "-t", "c", str(src), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and not out.exists() and r.stderr.strip() == "47 B => 42 B"
    r = subprocess.run([EXE, "-t", "d", str(tmp_path / "t.txt.et"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and not out.exists() and r.stderr.strip() == "38 B => 0 B"
    # -p prints the text (decode.zig:189)
    r = subprocess.run([EXE, "-pt", "d", str(tmp_path / "t.txt.et")], capture_output=True)
    assert r.stdout == text
    # -d: dictionary dump in DFS order (encode.zig:204-212), bits in output, time taken
    r = subprocess.run([EXE, "-dt", "c", str(src)], capture_output=True)
    lines = r.stdout.decode("utf-8", "replace").split("\n")
    _, _, order = O.build_dict(O.histogram(text))
    want = {"D": "00", "_": "01", "A": "10", "E": "110", "B": "1111", "\n": "11100", "C": "11101"}
    assert b"\nbits in output: 336\n" in r.stdout and b"time taken: " in r.stdout
    dump = r.stdout.split(b"\nbits in output")[0]
    codes = [seg.split(b"\n")[0].decode() for seg in dump.split(b" - ")[1:]]
    assert codes == [want[chr(s)] for s in order]
    # empty input: error.QueueEmpty, and the output file has already been created (main.zig:192)
    empty = tmp_path / "empty.txt"
    empty.write_bytes(b"")
    r = subprocess.run([EXE, "c", str(empty)], capture_output=True, text=True)
    assert r.returncode != 0 and (tmp_path / "empty.txt.et").exists() and (tmp_path / "empty.txt.et").stat().st_size == 0


def test_sharded_codec_over_rccl_world_size_1(ctx):
    """The N > 1 code path of entreepy_amd.sharded (histogram all_gather, offset plan,
    head shard, boundary all_gather, gather_file) through RCCL on the one GPU we have."""
    import torch
    import torch.distributed as dist

    from entreepy_amd import sharded
    from oracle import oracle as O

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        data = corpus.text_like(3_000_001, 41)
        text = torch.from_numpy(data).cuda()
        import entreepy_amd as E

        enc = torch.zeros(E.encode_bound(data.size) + 64, dtype=torch.uint8, device="cuda")
        codec = sharded.ShardedCodec(ctx, dist.group.WORLD, torch.device("cuda", 0))
        layout = codec.encode_shard(text, enc)
        image = codec.gather_file(enc, layout)
        assert image == O.encode(data)
        dec = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
        m = codec.decode_shard(enc, layout, dec)
        torch.cuda.synchronize()
        assert m == data.size and bool((dec[:m] == text).all())
    finally:
        dist.destroy_process_group()


def test_large_stream_properties(ctx):
    """Size-independent checks at a size the oracle does not finish in seconds
    (1 GiB text-like): decode(encode(x)) == x on the device, the image length equals
    header + ceil(sum(hist * len) / 8), and encoding the stream as 4 virtual shards at
    their bit offsets reproduces the single-stream body word for word."""
    import torch

    import entreepy_amd as E

    n = 1 << 30
    text = corpus.text_like_torch(n, 0x5EED0004, torch.device("cuda", 0))
    enc = torch.zeros(E.encode_bound(n) + 64, dtype=torch.uint8, device="cuda")
    et_len = ctx.encode_device(text, enc)
    dec = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    m = ctx.decode_device(enc[4:et_len], dec)
    torch.cuda.synchronize()
    assert m == n and torch.equal(dec[:n], text)
    hist = torch.bincount(text.view(torch.uint8).to(torch.int64), minlength=256).cpu().numpy().astype(np.uint64)
    cb = E.Codebook.from_histogram(hist)
    header = cb.header(n)
    assert enc[: len(header)].cpu().numpy().tobytes() == header
    assert et_len == len(header) + (cb.bits(hist) + 7) // 8
    # shards
    whole = enc.clone()  # zero-padded past et_len, so the last word can be compared whole
    bit = 8 * len(header)
    h = torch.zeros(256, dtype=torch.int64, device="cuda")
    for r in range(4):
        view = text[r * (n // 4) : (r + 1) * (n // 4)]
        ctx.histogram_device(view, h)
        out = torch.zeros(view.numel() + 64, dtype=torch.uint8, device="cuda")
        local = bit % 32
        end = ctx.encode_body_device(cb, view, out, local)
        torch.cuda.synchronize()
        w0, w1 = bit // 32, (bit + end - local + 31) // 32
        ref = whole[w0 * 4 : w1 * 4].clone()
        got = out[: (w1 - w0) * 4]
        # interior words identical; first/last word: the shard's bits are a subset of the stream's
        assert torch.equal(got[4:-4], ref[4:-4])
        assert bool(((got[:4] | ref[:4]) == ref[:4]).all()) and bool(((got[-4:] | ref[-4:]) == ref[-4:]).all())
        bit += end - local
    assert (bit + 7) // 8 == et_len


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_cold_decode_virtual_ranks(ranks, res_files):
    """et_decode_range_sync / _write: one .et stream cut into block ranges, each handled by
    its own et_ctx on the one GPU, exchange done by hand exactly as sharded.decode_cold
    does over RCCL.  Includes a forced wrong start to exercise the repair call."""
    import torch

    import entreepy_amd as E
    from oracle import oracle as O

    for data in (corpus.text_like(2_000_003, 61), np.frombuffer(res_files["a_midsummer_nights_dream.txt"], dtype=np.uint8), corpus.uniform(300_000, 9, 1, 256)):
        et = O.encode(data)
        comp = torch.from_numpy(np.frombuffer(et[4:], dtype=np.uint8).copy()).cuda()
        cb, n_symbols, body_off = E.parse_header(et[4:])
        ptr = comp.data_ptr() + body_off
        base_off, first_bit = body_off - (ptr & 3), (ptr & 3) * 8
        stream = comp[base_off:]
        n_blocks = (stream.numel() + 8191) // 8192
        ctxs, infos, spans = [], [], []
        for r in range(ranks):
            lo, hi = r * n_blocks // ranks, (r + 1) * n_blocks // ranks
            if hi == lo:
                continue
            c = E.Context(0)
            c.use_torch_stream()
            begin, end = lo * 8192, min(hi * 8192, stream.numel())
            start = first_bit if lo == 0 else -1
            if lo != 0 and r == 1:
                # deliberately wrong known start: must be repaired below
                start = 5 if begin else first_bit
            infos.append(c.decode_range_sync(cb, stream, begin, end, start))
            ctxs.append(c)
            spans.append((begin, end))
        for _ in range(ranks + 2):
            prev, wrong = first_bit, []
            for i, inf in enumerate(infos):
                if inf["start_bit"] != prev:
                    wrong.append((i, prev))
                prev = inf["exit_bit"]
            if not wrong:
                break
            for i, w in wrong:
                infos[i] = ctxs[i].decode_range_sync(cb, stream, spans[i][0], spans[i][1], w)
        else:
            raise AssertionError("did not settle")
        out, first = [], 0
        for c, inf in zip(ctxs, infos):
            take = max(0, min(inf["n_symbols"], n_symbols - first))
            buf = torch.empty(inf["n_symbols"] + 64, dtype=torch.uint8, device="cuda")
            m = c.decode_range_write(take, buf)
            torch.cuda.synchronize()
            out.append(buf[:m].cpu().numpy())
            first += inf["n_symbols"]
        assert np.concatenate(out).tobytes() == O.decode(et[4:])
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("tail_bytes", [1, 5, 15, 16])
def test_cold_decode_tiny_last_block(tail_bytes):
    """A stream that ends a few bytes into its last 8 KiB block, cut over as many virtual ranks
    as it has blocks: the block plan (sharded.cut_blocks) gives the tiny last block to the range
    before it; the concatenated pieces equal the oracle's decode of the (truncated) stream."""
    import torch

    import entreepy_amd as E
    from entreepy_amd.sharded import cut_blocks
    from oracle import oracle as O

    et = O.encode(corpus.text_like(120_000, 88))[4:]
    cb, n_symbols, body_off = E.parse_header(et)
    base = body_off - (body_off & 3)  # the tensor below is 4-byte aligned
    et = et[: base + 5 * 8192 + tail_bytes]
    comp = torch.from_numpy(np.frombuffer(et, dtype=np.uint8).copy()).cuda()
    assert (comp.data_ptr() + body_off) & 3 == body_off & 3
    first_bit = (body_off & 3) * 8
    stream = comp[base:]
    n_blocks = cut_blocks(stream.numel())
    assert n_blocks == (5 if tail_bytes < 16 else 6)
    ranks = n_blocks
    ctxs, infos = [], []
    for r in range(ranks):
        lo, hi = r * n_blocks // ranks, (r + 1) * n_blocks // ranks
        c = E.Context(0)
        c.use_torch_stream()
        begin, end = lo * 8192, (stream.numel() if hi == n_blocks else hi * 8192)
        infos.append(c.decode_range_sync(cb, stream, begin, end, first_bit if lo == 0 else -1))
        ctxs.append((c, begin, end))
    for _ in range(ranks + 2):
        prev, wrong = first_bit, []
        for i, inf in enumerate(infos):
            if inf["start_bit"] != prev:
                wrong.append((i, prev))
            prev = inf["exit_bit"]
        if not wrong:
            break
        for i, w in wrong:
            infos[i] = ctxs[i][0].decode_range_sync(cb, stream, ctxs[i][1], ctxs[i][2], w)
    else:
        raise AssertionError("did not settle")
    out, first = [], 0
    for (c, _, _), inf in zip(ctxs, infos):
        take = max(0, min(inf["n_symbols"], n_symbols - first))
        buf = torch.empty(inf["n_symbols"] + 64, dtype=torch.uint8, device="cuda")
        m = c.decode_range_write(take, buf)
        torch.cuda.synchronize()
        out.append(buf[:m].cpu().numpy())
        first += inf["n_symbols"]
        c.close()
    assert np.concatenate(out).tobytes() == O.decode(et)


@pytest.mark.parametrize("ranks", [2, 5])
def test_cold_decode_exhaustive_virtual_ranks(ranks):
    """et_decode_range_maps / _resolve: a stream of near-fixed-length codes (uniform alphabet:
    nothing to re-synchronise on) cut into block ranges, each on its own et_ctx; the 32-byte
    start->exit maps are chained by hand as sharded.decode_cold does after its all-gather."""
    import torch

    import entreepy_amd as E
    from oracle import oracle as O

    for data in (corpus.uniform(900_001, 31, 1, 256), corpus.uniform(500_000, 32, 10, 10 + 100)):
        # (255 symbols: a complete code of 7 and 8 bits -- the ranges' maps and starts by rows and columns, csrc/et_rowsync.hip;
        # 100 symbols: 6 and 7 bits -- the exit maps for every start offset, csrc/et_kernels_fallback.hip)
        by_rows = data.max() > 200
        et = O.encode(data)
        comp = torch.from_numpy(np.frombuffer(et[4:], dtype=np.uint8).copy()).cuda()
        cb, n_symbols, body_off = E.parse_header(et[4:])
        ptr = comp.data_ptr() + body_off
        base_off, first_bit = body_off - (ptr & 3), (ptr & 3) * 8
        stream = comp[base_off:]
        n_blocks = (stream.numel() + 8191) // 8192
        ctxs, maps, spans = [], [], []
        try:
            for r in range(ranks):
                lo, hi = r * n_blocks // ranks, (r + 1) * n_blocks // ranks
                c = E.Context(0)
                c.use_torch_stream()
                begin, end = lo * 8192, min(hi * 8192, stream.numel())
                m, n_starts = c.decode_range_maps(cb, stream, begin, end, first_bit if lo == 0 else -1)
                assert n_starts == cb.raw.max_length
                ctxs.append(c)
                maps.append(m)
                spans.append((begin, end))
            out, first, s_in = [], 0, first_bit
            for c, m in zip(ctxs, maps):
                inf = c.decode_range_resolve(s_in)
                assert inf["start_bit"] == s_in and inf["exit_bit"] == m[s_in] and inf["row_walk"] == by_rows
                s_in = m[s_in]
                take = max(0, min(inf["n_symbols"], n_symbols - first))
                buf = torch.empty(inf["n_symbols"] + 64, dtype=torch.uint8, device="cuda")
                k = c.decode_range_write(take, buf)
                torch.cuda.synchronize()
                out.append(buf[:k].cpu().numpy())
                first += inf["n_symbols"]
            assert np.concatenate(out).tobytes() == O.decode(et[4:])
        finally:
            for c in ctxs:
                c.close()


def test_cold_decode_over_rccl_world_size_1(ctx):
    import torch
    import torch.distributed as dist

    from entreepy_amd import sharded
    from oracle import oracle as O

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        # text: run-in synchronisation; uniform 200-symbol stream: the exhaustive map exchange
        for data in (corpus.text_like(1_500_000, 71), corpus.uniform(700_000, 72, 1, 201)):
            et = O.encode(data)
            comp = torch.from_numpy(np.frombuffer(et[4:], dtype=np.uint8).copy()).cuda()
            dec = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
            codec = sharded.ShardedCodec(ctx, dist.group.WORLD, torch.device("cuda", 0))
            m, first = codec.decode_cold(comp, dec)
            torch.cuda.synchronize()
            assert first == 0 and m == data.size and dec[:m].cpu().numpy().tobytes() == data.tobytes()
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------- file pipeline
def _file_ctx(chunk_mb, threads):
    """A context whose staging pipeline uses small chunks, so that a few MiB span many."""
    import os

    import entreepy_amd as E

    old = {k: os.environ.get(k) for k in ("ET_IO_CHUNK_MB", "ET_IO_THREADS")}
    os.environ["ET_IO_CHUNK_MB"], os.environ["ET_IO_THREADS"] = str(chunk_mb), str(threads)
    try:
        c = E.Context(0)
        c.encode(b"ab")  # the pipeline reads its knobs on first use
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return c


@pytest.mark.parametrize("n,chunk_mb,threads", [(1, 1, 1), (4096, 1, 3), ((1 << 20) - 1, 1, 4), ((1 << 20) + 1, 1, 4), (5 * (1 << 20) + 12345, 1, 5), (3 * (1 << 20), 2, 8)])
def test_file_pipeline_matches_oracle(tmp_path, n, chunk_mb, threads):
    """et_encode_fd / et_decode_fd: file -> pinned chunks -> HBM -> pinned chunks -> file, with
    chunk boundaries inside the stream; the .et file is the oracle's, the round trip exact."""
    from oracle import oracle as O

    c = _file_ctx(chunk_mb, threads)
    data = corpus.text_like(n, 77 + n % 13).tobytes() if n > 1 else b"x"
    src, et, back = tmp_path / "in.txt", tmp_path / "out.et", tmp_path / "back.txt"
    src.write_bytes(data)
    assert c.encode_file(str(src), str(et)) == (n, len(O.encode(data)))
    assert et.read_bytes() == O.encode(data)
    consumed, produced = c.decode_file(str(et), str(back))
    assert consumed == et.stat().st_size - 4
    want = O.decode(O.encode(data)[4:])
    assert produced == len(want) and back.read_bytes() == want
    # host-pointer calls run through the same staging buffers
    assert c.encode(data) == O.encode(data) and c.decode(O.encode(data)[4:]) == want
    # dry run: coded, nothing written
    assert c.encode_file(str(src), None) == (n, len(O.encode(data)))


def test_file_pipeline_errors(tmp_path):
    import entreepy_amd as E

    c = E.Context(0)
    empty = tmp_path / "empty"
    empty.write_bytes(b"")
    with pytest.raises(E.EmptyInputError):  # error.QueueEmpty, as encode() on an empty slice
        c.encode_file(str(empty), str(tmp_path / "o.et"))
    short = tmp_path / "short.et"
    short.write_bytes(b"\xe7\xc0\xde")
    with pytest.raises(E.EntreepyError):
        c.decode_file(str(short), str(tmp_path / "o.txt"))
    junk = tmp_path / "junk.et"
    junk.write_bytes(b"\xe7\xc0\xde\x01" + bytes(range(1, 200)))
    with pytest.raises(E.EntreepyError):
        c.decode_file(str(junk), str(tmp_path / "o2.txt"))


def test_cli_refuses_files_without_the_magic(tmp_path, res_files):
    """main.zig:199 leaves validation as a TODO and would decode whatever follows byte 4;
    the CLI refuses such files (and leaves the created output empty, as a failed run of the
    reference does, main.zig:192)."""
    text = next(iter(res_files.values()))
    bad = tmp_path / "not_et.et"
    bad.write_bytes(text)
    out = tmp_path / "o.txt"
    r = subprocess.run([EXE, "d", str(bad), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "magic" in r.stderr and out.stat().st_size == 0
    tiny = tmp_path / "tiny.et"
    tiny.write_bytes(b"\xe7\xc0")
    r = subprocess.run([EXE, "d", str(tiny), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 1


def test_bench_line_has_the_contract_keys():
    """bench.py as the driver runs it (smaller stream, fewer steps): one JSON line on stdout with the contract's keys,
    the roofline and CPU-baseline objects, a kernel table that names what a decode really ran, and the second workload."""
    import json
    import sys

    env = dict(os.environ, ET_BENCH_BYTES=str(32 << 20))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--cpu-threads", "4",
                        "--workloads", "enwik-like,text-5M"],  # (the 4 GiB and 16 GiB streams of BASELINE config 5 are the default run's)
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["unit"] == "GB/s" and d["dtype"] == "u8" and d["value"] > 0
    assert abs(d["value"] - (32 << 20) / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["kernel"] in d["kernels"] and "k_tw_sync" in d["kernels"] and "k_dec_write_wave" in d["kernels"]
    # the dominant kernel's duration comes from the events it carries through the TIMED region (the other phases' from the
    # instrumented set-up steps, and the line says so)
    assert rf["kernel"] != "k_dec_write_wave" or "timed region" in rf["measured"]
    assert "phase_ms_measured" in d and all(d["phase_ms"][k] > 0 for k in ("hist", "enc_body", "dec_sync_first", "dec_body"))
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    assert "error" not in d["workloads"]["enwik-like"] and d["workloads"]["enwik-like"]["round_trip_GBps"] > 0
    t5 = d["workloads"]["text-5M"]
    assert t5["bytes"] == 5_458_199 and t5["verified"] is True and t5["round_trip_GBps"] > 0 and "k_tw_sync" in t5["decode_path"]
    # what the line does NOT measure itself says so: the PMC traffic is the committed passes' (and absent for a run of another size);
    # whether the round trips were compared, and which build of the library ran
    assert d["verified"] is True and d["library"].endswith("libentreepy_hip.so")
    assert rf["traffic"] is None and "1073741824" in rf["traffic_source"]
    # the run as the contract words it (W warm-up steps, K timed) is in the line beside the steady-state figure
    assert d["value_cold"] > 0 and abs(d["value_cold"] - (32 << 20) / (d["ms_per_step_cold"] * 1e-3) / 1e9) < 0.01 * d["value_cold"]
    assert d["config"]["decode"].startswith("cold")


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher on the command line and no WORLD_SIZE in the environment: bench.py starts
    the two ranks itself (torch.distributed.run as a child process), relays rank 0's one line and exits 0.  Rehearsed on the
    box's one GPU: both ranks on cuda:0, rows exchanged over gloo through the library's exchange callback."""
    import json
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ET_BENCH_BYTES=str(16 << 20), ET_BENCH_DEVICE="0", ET_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--ref-value", "100"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * (16 << 20) / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]
    for k in ("exchange_ms", "seam_ms", "concat_ms", "scaling_efficiency", "exchange"):
        assert k in d, k
    assert abs(d["scaling_efficiency"] - d["value"] / 200.0) < 1e-3 and d["scaling_efficiency_against"] == "--ref-value"
    assert d["config"]["decode"].startswith("shard ranges with the encode's offsets")
    # what an efficiency is computed from when nobody hands in --ref-value: the same step on one GPU per rank, this run; and the
    # cold decode of the concatenated image by the ranks, beside it
    assert d["value_n1_same_step"] > 0 and abs(d["value_n1_same_step"] - (16 << 20) / (d["ms_per_step_n1_same_step"] * 1e-3) / 1e9) < 0.01 * d["value_n1_same_step"]
    assert isinstance(d["decode_cold_sharded_ms"], float) and d["decode_cold_sharded_ms"] > 0, d["decode_cold_sharded_ms"]


def test_bench_n1_shard_style_decode():
    """`bench.py --decode shard` at N = 1: the step an N > 1 value is comparable with (the body decoded with the encode's own
    code table, no header hand-over); the line says which decode it timed."""
    import json
    import sys

    env = dict(os.environ, ET_BENCH_BYTES=str(16 << 20))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--decode", "shard", "--no-cpu-baseline", "--no-extra-workloads"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["verified"] is True
    assert d["config"]["decode"].startswith("shard ranges with the encode's offsets")

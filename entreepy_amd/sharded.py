"""One stream sharded by contiguous chunk across the GPUs of a node (one process per GPU).

Encode has exactly one exchange step: the byte histogram.  Every rank
  1. histograms its chunk on its GPU (K1),
  2. all-gathers the 256 x u64 local histograms (RCCL over xGMI; 2 KiB per rank --
     latency-bound, one collective).  The sum is the global histogram the reference
     computes in encode.zig:43-47; the individual rows give every shard's body bit
     count as sum(hist_r * length) with no further data pass or collective,
  3. builds the same code table from the same global histogram (deterministic host
     code, et_build_codebook), so no table broadcast is needed,
  4. packs its chunk at its global bit offset (K2 + K4).  Rank r's piece covers file
     words [S_r // 32, ceil(S_{r+1} / 32)) and holds zeros for the bits of a shared
     boundary word that belong to a neighbour.
The file then exists as per-rank pieces resident in HBM; the bit-offset-adjusted
concatenation (`gather_file`: pieces OR-ed into place on rank 0) is used by tests and
the CLI and is not part of the timed step (a host `pwrite` per shard needs no gather).

The collectives go through torch.distributed (backend "nccl" == RCCL on ROCm; "gloo"
in the CPU tests, where a fake compute backend stands in for the GPU).
"""
import time

import numpy as np
import torch
import torch.distributed as dist

from .codec import Codebook


def plan_shards(hists):
    """hists: uint64 [world, 256] local histograms ->
    (codebook, header bytes, start_bits[world + 1]) with start_bits measured from bit 0
    of the FILE (header included).  One library call (et_plan_shards): sum of the rows ->
    code table (encode.zig:54-214) -> header -> every shard's body bits = sum of count x code
    length.  Raises EmptyInputError when every shard is empty (encode.zig:137-138)."""
    import ctypes

    from . import _native as N
    from .codec import _check

    hists = np.ascontiguousarray(hists, dtype=np.uint64)
    world = hists.shape[0]
    cb = Codebook()
    header = np.empty(8192, dtype=np.uint8)
    starts = np.empty(world + 1, dtype=np.uint64)
    n = ctypes.c_size_t(0)
    _check(N.lib().et_plan_shards(hists.ctypes.data, world, ctypes.byref(cb.raw), header.ctypes.data, header.size, ctypes.byref(n), starts.ctypes.data))
    return cb, header[: n.value].tobytes(), [int(x) for x in starts]


def piece_words(starts, r):
    """File words [lo, hi) held by rank r's local buffer."""
    lo = 0 if r == 0 else starts[r] // 32
    hi = (starts[r + 1] + 31) // 32
    return lo, max(hi, lo)


def owned_words(starts, r):
    """File words rank r contributes to the concatenation: a word shared by several
    ranks belongs to the first of them."""
    lo = 0 if r == 0 else (starts[r] + 31) // 32
    hi = (starts[r + 1] + 31) // 32
    return lo, max(hi, lo)


def cut_blocks(stream_bytes):
    """8 KiB blocks a cold stream is cut over.  et_decode_range_sync wants >= 16 bytes of stream
    after every range but the last (the run-out of a code that starts at the range's end), so a
    last block shorter than that is not a block of its own: the range that ends the stream takes it."""
    n_blocks = (stream_bytes + 8191) // 8192
    if n_blocks > 1 and stream_bytes - (n_blocks - 1) * 8192 < 16:
        n_blocks -= 1
    return n_blocks


class ShardedCodec:
    def __init__(self, ctx, group, device):
        self.ctx = ctx
        self.group = group
        self.device = device
        self.world = dist.get_world_size(group) if group is not None else 1
        self.rank = dist.get_rank(group) if group is not None else 0
        # collectives run on device tensors with RCCL ("nccl"); with gloo (CPU tests, and
        # the 2-ranks-on-one-GPU rehearsal of bench.py) they run on host copies
        self.coll_device = device if (group is None or dist.get_backend(group) == "nccl") else torch.device("cpu")
        self.hist = torch.zeros(256, dtype=torch.int64, device=device)
        self.all_hists = torch.zeros(self.world * 256, dtype=torch.int64, device=self.coll_device)
        # the gathered counts come to the host through a pinned buffer (a pageable .cpu() costs ~10 us more per step)
        self.h_hists = torch.zeros(self.world * 256, dtype=torch.int64).pin_memory() if self.coll_device.type == "cuda" else None
        self._hdr_len = {}

    # ------------------------------------------------------------------ encode
    def encode_shard(self, text, enc, timings=True):
        """text: this rank's chunk (uint8 device tensor); enc: uint8 device buffer of
        encode_bound(len) + 64 bytes, 16-byte aligned.  Returns the layout dict
        decode_shard / gather_file need.  timings=False (one GPU): do not wait for the
        phase timings here ("timings" is None; fetch them later with
        single_encode_timings(), e.g. once the decode has been enqueued)."""
        n = text.numel()
        ctx = self.ctx
        if self.group is None:
            et_len = ctx.encode_device(text, enc)
            hdr = self._hdr_len.get(et_len)
            if hdr is None:
                cbk, _, off = _parse_device_header(enc, et_len)
                hdr = self._hdr_len[et_len] = off + 4
            return {"world": 1, "single": True, "n": n, "et_len": et_len, "header_len": hdr, "body_bytes": et_len - hdr,
                    "timings": self.single_encode_timings() if timings else None}

        ctx.histogram_device(text, self.hist)
        t_x0 = time.perf_counter()
        dist.all_gather_into_tensor(self.all_hists, self.hist.to(self.coll_device), group=self.group)
        if self.h_hists is not None:
            self.h_hists.copy_(self.all_hists, non_blocking=True)
            torch.cuda.current_stream(self.coll_device).synchronize()
            hists = self.h_hists.numpy().view(np.uint64).reshape(self.world, 256).copy()
        else:
            hists = self.all_hists.view(self.world, 256).numpy().astype(np.uint64)
        t_x1 = time.perf_counter()
        cb, header, starts = plan_shards(hists)
        t_h1 = time.perf_counter()
        r = self.rank
        if n and hasattr(ctx, "histogram_on_host"):
            ctx.histogram_on_host(hists[r])  # this rank's row of the exchange: no second read-back in the shard encode
        if r == 0:
            end = ctx.encode_head_shard_device(cb, text, enc, header)
            local_start = starts[0]
        else:
            local_start = starts[r] % 32
            end = ctx.encode_body_device(cb, text, enc, local_start)
        assert end - local_start == starts[r + 1] - starts[r]
        # host-side figures now; the GPU phases are read from the context's events on demand
        self._host_timings = {"enc_host": (t_h1 - t_x1) * 1e3, "exchange": (t_x1 - t_x0) * 1e3}
        return {"world": self.world, "single": False, "n": n, "codebook": cb, "header_len": len(header) if r == 0 else 0, "starts": starts,
                "local_start_bit": local_start, "end_bit": end, "body_bytes": (starts[r + 1] - starts[r] + 7) // 8,
                "timings": self.encode_timings() if timings else None}

    def encode_timings(self):
        """Phase timings (ms) of the last encode_shard; waits for its last kernel.  With a
        group, "exchange" is the wall clock of the histogram all-gather including the wait
        for K1 before it (host clock; on the GPU's clock it lies inside enc_scan), and
        enc_total = hist + enc_scan + enc_body."""
        if self.group is None:
            return self.single_encode_timings()
        t = self.ctx.timings("encode")
        out = {"hist": t["hist_ms"], "enc_scan": t["scan_ms"], "enc_body": t["body_ms"], **self._host_timings}
        # scan_ms spans everything between K1 and K4 on the GPU's clock: histogram reduce, the exchange
        # and the host's code construction (also listed on their own, host clock), tile scan, uploads
        out["enc_total"] = out["hist"] + out["enc_scan"] + out["enc_body"]
        return out

    def single_encode_timings(self):
        t = self.ctx.timings("encode")
        return {"hist": t["hist_ms"], "enc_host": t["host_ms"], "enc_scan": t["scan_ms"], "enc_body": t["body_ms"],
                "enc_total": t["total_ms"], "exchange": 0.0}

    # ------------------------------------------------------------------ decode
    def decode_shard(self, enc, layout, dec):
        """Decode this rank's piece back into dec; returns the symbol count.  With one
        GPU this is the reference's decode(file[4..]) (header parsed from the stream);
        with several, every rank decodes its own bit range [S_r, S_{r+1}) using the
        offsets the encode step produced (an in-memory pipeline, not a cold .et read)."""
        ctx = self.ctx
        if layout["single"]:
            return ctx.decode_device(enc[4 : layout["et_len"]], dec)
        start = layout["local_start_bit"]
        byte0 = start // 8
        byte1 = (layout["end_bit"] + 7) // 8
        return ctx.decode_body_device(layout["codebook"], enc[byte0:byte1], layout["n"], dec, start % 8)

    # ------------------------------------------------------------- cold decode
    def decode_cold(self, compressed, dec):
        """Decode ONE .et stream (uint8 device tensor holding the file minus its first 4
        bytes -- decode.zig's `compressed_text` -- on every rank, or at least this rank's
        block range with 16 bytes on either side) across the ranks of the group, with no
        side information.  The body is cut at multiples of 8 KiB from its 4-byte aligned
        base; every rank synchronises its range (running in from the 16 bytes before it),
        the ranks all-gather (start, exit, symbols), a rank whose start is not its
        predecessor's exit repairs, and when all agree each rank writes its symbols to
        `dec`.  Returns (symbols written by this rank, global index of its first symbol).
        Expected rounds: 1 (a run-in is right 99.6 % of the time on text)."""
        from .codec import parse_header

        world, r = self.world, self.rank
        head = compressed[: min(compressed.numel(), 8192)].cpu().numpy()
        cb, n_symbols, body_off = parse_header(head)
        ptr = compressed.data_ptr() + body_off
        base_off = body_off - (ptr & 3)          # 4-byte aligned base of the body inside `compressed`
        first_bit = (ptr & 3) * 8
        stream = compressed[base_off:]
        n_blocks = cut_blocks(stream.numel())
        lo_b, hi_b = r * n_blocks // world, (r + 1) * n_blocks // world
        begin, end = lo_b * 8192, (stream.numel() if hi_b == n_blocks else hi_b * 8192)
        active = hi_b > lo_b and cb.raw.n_coded > 0 and n_symbols > 0
        info = {"start_bit": 0, "exit_bit": 0, "n_symbols": 0}
        # Near-fixed-length codes do not self-synchronise: run-ins find nothing.  Every rank then
        # computes its range's exit for each possible start (et_decode_range_maps), the 32-byte
        # maps are all-gathered and chained from the stream's start, and each rank resolves its
        # range with the start that reaches it -- one exchange, no repair rounds.
        exhaustive = cb.raw.n_coded > 2 and cb.raw.max_length <= cb.raw.min_length + 1
        if exhaustive:
            mine_map = torch.arange(32, dtype=torch.uint8)  # a rank without blocks passes the start on
            if active:
                m, _ = self.ctx.decode_range_maps(cb, stream, begin, end, first_bit if lo_b == 0 else -1)
                mine_map = torch.tensor(list(m), dtype=torch.uint8)
            all_maps = torch.zeros(world * 32, dtype=torch.uint8, device=self.coll_device)
            dist.all_gather_into_tensor(all_maps, mine_map.to(self.coll_device), group=self.group)
            maps = all_maps.view(world, 32).cpu().numpy()
            s_in = first_bit
            for q in range(r):
                s_in = int(maps[q, s_in])
            if active:
                info = self.ctx.decode_range_resolve(s_in)
        elif active:
            info = self.ctx.decode_range_sync(cb, stream, begin, end, first_bit if lo_b == 0 else -1)
        table = torch.zeros(world * 3, dtype=torch.int64, device=self.coll_device)
        mine = torch.zeros(3, dtype=torch.int64, device=self.coll_device)
        rounds = 0
        while True:
            rounds += 1
            mine[0], mine[1], mine[2] = (info["start_bit"] if active else -1), (info["exit_bit"] if active else -1), info["n_symbols"]
            dist.all_gather_into_tensor(table, mine, group=self.group)
            t = table.view(world, 3).cpu().numpy()
            # the exit that reaches rank q: the nearest active predecessor's (inactive ranks hold no blocks)
            want = {}
            prev_exit = first_bit
            for q in range(world):
                if t[q, 0] >= 0:
                    want[q] = prev_exit
                    prev_exit = int(t[q, 1])
            wrong = [q for q in want if want[q] != int(t[q, 0])]
            if not wrong:
                break
            if r in wrong:
                info = self.ctx.decode_range_sync(cb, stream, begin, end, want[r])
            assert rounds <= world + 1, "cold decode did not settle"
        counts = t[:, 2]
        first = int(counts[:r].sum())
        take = max(0, min(int(counts[r]), n_symbols - first))
        written = self.ctx.decode_range_write(take, dec) if active and take else 0
        return written, first

    # ------------------------------------------------------------------ concat
    def gather_file(self, enc, layout):
        """Bit-offset-adjusted concatenation on rank 0 -> bytes (None elsewhere)."""
        if layout["single"]:
            return enc[: layout["et_len"]].cpu().numpy().tobytes()
        starts, r = layout["starts"], self.rank
        lo, hi = piece_words(starts, r)
        mine = enc[: (hi - lo) * 4].cpu().numpy().tobytes()
        pieces = [None] * self.world
        dist.all_gather_object(pieces, mine, group=self.group)
        if r != 0:
            return None
        # A word two (or more) neighbouring pieces share holds disjoint bits of each and
        # zeros elsewhere: OR the pieces into place.
        image = np.zeros(((starts[-1] + 31) // 32) * 4, dtype=np.uint8)
        for q, piece in enumerate(pieces):
            qlo, _ = piece_words(starts, q)
            a = np.frombuffer(piece, dtype=np.uint8)
            image[qlo * 4 : qlo * 4 + a.size] |= a
        return image[: (starts[-1] + 7) // 8].tobytes()


def _parse_device_header(enc, et_len):
    from .codec import parse_header

    head = enc[4 : min(et_len, 8192)].cpu().numpy()
    return parse_header(head)

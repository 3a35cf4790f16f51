"""One stream sharded by contiguous chunk across the GPUs of a node (one process per GPU).

The sequencing lives behind the C ABI (csrc/et_sharded.cpp: et_encode_sharded,
et_shard_merge_seams, et_shard_write_fd / _place / _gather, et_decode_sharded); this module is
its torch.distributed host: it supplies the exchange -- RCCL over xGMI for backend "nccl" (the
library opens its own communicator from an id broadcast once), an all-gather over the process
group otherwise (gloo) -- and keeps the layout dicts bench.py and the tests use.

Encode has exactly one exchange step: the byte histogram.  Every rank
  1. histograms its chunk on its GPU (K1),
  2. all-gathers the 256 x u64 local histograms (2 KiB per rank -- latency-bound, one
     collective).  The sum is the global histogram the reference computes in
     encode.zig:43-47; the individual rows give every shard's body bit count as
     sum(hist_r * length) with no further data pass or collective,
  3. builds the same code table from the same global histogram (deterministic host code,
     et_plan_shards), so no table broadcast is needed,
  4. packs its chunk at its global bit offset (K2 + K4).  Rank r's piece covers file words
     [S_r // 32, ceil(S_{r+1} / 32)) and holds zeros for the bits of a shared boundary word
     that belong to a neighbour.
The bit-offset-adjusted concatenation (encode.zig:319 writes ONE image): the owner of a shared
word -- the first shard in it -- receives its successors' bits (merge_seams: one 8-byte
exchange), after which every rank's owned words are a disjoint range of the image and go to a
file (pwrite per shard) or to rank 0's image over xGMI.

A context that is not an entreepy_amd.Context (the CPU tests' oracle-backed stand-in for the
GPU) takes the same steps in Python; the word arithmetic is the library's either way
(et_plan_shards, et_shard_words, et_seam_word).
"""
import time

import numpy as np
import torch
import torch.distributed as dist

from .codec import Codebook, Context, Group, seam_word, shard_words


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
    return shard_words(starts, len(starts) - 1, r)[:2]


def owned_words(starts, r):
    """File words rank r contributes to the concatenation: a word shared by several
    ranks belongs to the first of them."""
    return shard_words(starts, len(starts) - 1, r)[2:]


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
        # A real Context sequences through the library (et_sharded.cpp); the exchange is RCCL of the
        # library's own when the process group is RCCL, else an all-gather over the process group.
        self.lib_group = None
        if isinstance(ctx, Context):
            ctx.use_torch_stream()  # the tensors handed in are produced (and consumed) on torch's current stream
        if isinstance(ctx, Context) and group is not None:
            if dist.get_backend(group) == "nccl":
                ident = [Group.rccl_unique_id() if self.rank == 0 else None]
                dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0), group=group)
                self.lib_group = Group(ctx, self.rank, self.world, rccl_id=ident[0])
            else:
                self.lib_group = Group(ctx, self.rank, self.world, allgather=self._gather_bytes)

    def _gather_bytes(self, mine):
        """The exchange callback over the process group: bytes of this rank -> bytes of all, rank order."""
        send = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        recv = torch.empty(self.world * send.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv.numpy().tobytes()

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
            # header length from the call's own code table (9 bytes + the bit-packed dictionary, encode.zig:259-299)
            hdr = len(ctx.last_codebook().header(n))
            return {"world": 1, "single": True, "n": n, "et_len": et_len, "header_len": hdr, "body_bytes": et_len - hdr,
                    "timings": self.single_encode_timings() if timings else None}
        if self.lib_group is not None:
            i = self.lib_group.encode_sharded(text, enc)
            self._host_timings = {"enc_host": i["plan_ms"], "exchange": i["exchange_ms"]}
            return {"world": self.world, "single": False, "n": n, "codebook": self.lib_group.codebook(), "header_len": i["header_len"],
                    "starts": self.lib_group.start_bits(), "local_start_bit": i["local_start_bit"],
                    "end_bit": i["local_start_bit"] + (i["end_bit"] - i["start_bit"]), "body_bytes": (i["end_bit"] - i["start_bit"] + 7) // 8,
                    "info": i, "timings": self.encode_timings() if timings else None}

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

        if self.lib_group is not None:
            return self.lib_group.decode_sharded(compressed, dec)
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
    def merge_seams(self, enc, layout):
        """The owner of a word several shards share receives the later shards' bits (in place, in enc)."""
        if layout["single"]:
            return
        if self.lib_group is not None:
            self.lib_group.merge_seams(enc)
            return
        starts, r = layout["starts"], self.rank
        lo, hi = piece_words(starts, r)
        holds = starts[r + 1] > starts[r] or r == 0
        words = enc[: (hi - lo) * 4].cpu().numpy().view(np.uint32) if holds and hi > lo else np.zeros(0, dtype=np.uint32)
        mine = torch.tensor([int(words[0]), int(words[-1])] if words.size else [0, 0], dtype=torch.int64)
        both = torch.zeros(2 * self.world, dtype=torch.int64)
        dist.all_gather_into_tensor(both, mine, group=self.group)
        merged = seam_word(starts, self.world, r, both.numpy().astype(np.uint32))
        if merged is not None:
            enc[(hi - lo - 1) * 4 : (hi - lo) * 4] = torch.from_numpy(np.array([merged], dtype=np.uint32).view(np.uint8).copy()).to(enc.device)

    def concat_on_rank0(self, enc, layout):
        """The image as a device tensor on rank 0 (None elsewhere): seam merge + owned words over xGMI (RCCL
        groups) -- what bench.py times as concat_ms.  Without RCCL: through gather_file's host path."""
        if layout["single"]:
            return enc[: layout["et_len"]]
        if self.lib_group is not None and dist.get_backend(self.group) == "nccl":
            file_bytes = (layout["starts"][-1] + 7) // 8
            self.merge_seams(enc, layout)
            image = torch.empty((file_bytes + 3) // 4 * 4, dtype=torch.uint8, device=enc.device) if self.rank == 0 else None
            self.lib_group.gather(enc, image, 0)
            return image[:file_bytes] if self.rank == 0 else None
        data = self.gather_file(enc, layout)
        return torch.frombuffer(bytearray(data), dtype=torch.uint8).to(enc.device) if data is not None else None

    def gather_file(self, enc, layout):
        """Bit-offset-adjusted concatenation on rank 0 -> bytes (None elsewhere): seams merged, then
        every rank's owned words into place -- over xGMI (RCCL send/recv) into an image on rank 0's
        GPU, or, without RCCL, as one tensor gather over the process group."""
        if layout["single"]:
            return enc[: layout["et_len"]].cpu().numpy().tobytes()
        starts, r = layout["starts"], self.rank
        file_bytes = (starts[-1] + 7) // 8
        self.merge_seams(enc, layout)
        if self.lib_group is not None and dist.get_backend(self.group) == "nccl":
            image = torch.zeros((file_bytes + 3) // 4 * 4, dtype=torch.uint8, device=enc.device) if r == 0 else None
            self.lib_group.gather(enc, image, 0)
            return image[:file_bytes].cpu().numpy().tobytes() if r == 0 else None
        plo, _, olo, ohi = shard_words(starts, self.world, r)
        spans = [shard_words(starts, self.world, q)[2:] for q in range(self.world)]
        longest = max(1, max(h - l for l, h in spans) * 4)
        mine = torch.zeros(longest, dtype=torch.uint8)
        mine[: (ohi - olo) * 4] = enc[(olo - plo) * 4 : (ohi - plo) * 4].cpu()
        pieces = [torch.zeros(longest, dtype=torch.uint8) for _ in range(self.world)] if r == 0 else None
        dist.gather(mine, pieces, dst=dist.get_global_rank(self.group, 0), group=self.group)
        if r != 0:
            return None
        image = np.zeros(((starts[-1] + 31) // 32) * 4, dtype=np.uint8)
        for (l, h), piece in zip(spans, pieces):
            image[l * 4 : h * 4] = piece.numpy()[: (h - l) * 4]
        return image[:file_bytes].tobytes()


def _parse_device_header(enc, et_len):
    from .codec import parse_header

    head = enc[4 : min(et_len, 8192)].cpu().numpy()
    return parse_header(head)

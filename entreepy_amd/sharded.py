"""One stream sharded by contiguous chunk across the GPUs of a node (one process per GPU).

The sequencing lives behind the C ABI (csrc/et_shard_seq.cpp: et_encode_sharded,
et_shard_merge_seams, et_shard_write_fd / _place / _gather, et_decode_sharded); this module is
its torch.distributed host: it supplies the exchange -- RCCL over xGMI for backend "nccl" (the
library opens its own communicator from an id broadcast once), an all-gather over the process
group otherwise (gloo) -- and keeps the layout dicts bench.py and the tests use.  There is no
second implementation of the sequence here: the CPU tests hand in a Group whose library is a
build of the same et_shard_seq.cpp over a stand-in for the GPU (tests/support).

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
"""
import numpy as np
import torch
import torch.distributed as dist

from .codec import Codebook, Context, Group, shard_words


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


class _SingleLayout(dict):
    """Layout of a one-GPU encode.  "header_len" / "body_bytes" are worked out when somebody asks (from the call's own code
    table: 9 bytes + the bit-packed dictionary, encode.zig:259-299) -- not on the path between an encode and the decode behind it."""

    def __init__(self, ctx, **kw):
        super().__init__(**kw)
        self._ctx = ctx

    def __missing__(self, key):
        if key not in ("header_len", "body_bytes"):
            raise KeyError(key)
        hdr = len(self._ctx.last_codebook().header(self["n"]))
        self["header_len"], self["body_bytes"] = hdr, self["et_len"] - hdr
        return self[key]


class ShardedCodec:
    """ctx: this rank's Context; group: the torch.distributed process group (None: one GPU, no group);
    lib_group: a codec.Group to sequence through instead of one made here, or a callable(codec) that makes one (tests:
    a Group over the CPU stand-in whose exchange is this codec's gather_bytes)."""

    def __init__(self, ctx, group, device, lib_group=None):
        self.ctx = ctx
        self.group = group
        self.device = device
        self.world = dist.get_world_size(group) if group is not None else 1
        self.rank = dist.get_rank(group) if group is not None else 0
        self.lib_group = None
        self.rccl = False
        if isinstance(ctx, Context):
            ctx.use_torch_stream()  # the tensors handed in are produced (and consumed) on torch's current stream
        if lib_group is not None:
            self.lib_group = lib_group(self) if callable(lib_group) else lib_group
        elif group is not None:
            # The exchange: RCCL of the library's own when the process group is RCCL, else an all-gather over the process group.
            if dist.get_backend(group) == "nccl":
                ident = [Group.rccl_unique_id() if self.rank == 0 else None]
                dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0), group=group)
                self.lib_group = Group(ctx, self.rank, self.world, rccl_id=ident[0])
                self.rccl = True
            else:
                self.lib_group = Group(ctx, self.rank, self.world, allgather=self.gather_bytes)

    def gather_bytes(self, mine):
        """The exchange callback over the process group: bytes of this rank -> bytes of all, rank order."""
        send = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        recv = torch.empty(self.world * send.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv.numpy().tobytes()

    # ------------------------------------------------------------------ encode
    def encode_shard(self, text, enc, timings=True):
        """text: this rank's chunk (uint8 device tensor); enc: uint8 device buffer of
        encode_bound(len) + 64 bytes, 16-byte aligned.  Returns the layout dict
        decode_shard / gather_file need.  timings=False: do not wait for the
        phase timings here ("timings" is None; fetch them later with
        encode_timings(), e.g. once the decode has been enqueued)."""
        n = text.numel()
        ctx = self.ctx
        if self.group is None:
            et_len = ctx.encode_device(text, enc)
            return _SingleLayout(ctx, world=1, single=True, n=n, et_len=et_len, timings=self.single_encode_timings() if timings else None)
        i = self.lib_group.encode_sharded(text, enc)
        self._host_timings = {"enc_host": i["plan_ms"], "exchange": i["exchange_ms"]}
        return {"world": self.world, "single": False, "n": n, "codebook": self.lib_group.codebook(), "header_len": i["header_len"],
                "starts": self.lib_group.start_bits(), "local_start_bit": i["local_start_bit"],
                "end_bit": i["local_start_bit"] + (i["end_bit"] - i["start_bit"]), "body_bytes": (i["end_bit"] - i["start_bit"] + 7) // 8,
                "info": i, "timings": self.encode_timings() if timings else None}

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
    DECODE_SINGLE = "cold: decode(file[4..]), header and dictionary parsed from the stream (decode.zig:13)"
    DECODE_SHARD = ("shard ranges with the encode's offsets: every rank decodes the bits of its own shard with the code table and start bit of "
                    "the encode step that produced them (an in-memory pipeline; no header hand-over or parse, no exchange)")

    def decode_shard(self, enc, layout, dec, as_shard=False):
        """Decode this rank's piece back into dec; returns the symbol count.  With one
        GPU this is the reference's decode(file[4..]) (header parsed from the stream);
        with several, every rank decodes its own bit range [S_r, S_{r+1}) using the
        offsets the encode step produced (an in-memory pipeline, not a cold .et read).
        as_shard: one GPU, decoded the way the ranks of a group decode their shards -- the body with the encode's own
        code table, no header hand-over (what bench.py compares an N > 1 step with)."""
        ctx = self.ctx
        if layout["single"] and as_shard:
            hdr = layout["header_len"]
            return ctx.decode_body_device(ctx.last_codebook(), enc[hdr : layout["et_len"]], layout["n"], dec, 0)
        if layout["single"]:
            return ctx.decode_device(enc, dec, 4, layout["et_len"] - 4)  # main.zig:204: text_in[4..]
        start = layout["local_start_bit"]
        byte0 = start // 8
        byte1 = (layout["end_bit"] + 7) // 8
        return ctx.decode_body_device(layout["codebook"], enc[byte0:byte1], layout["n"], dec, start % 8)

    # ------------------------------------------------------------- cold decode
    def decode_cold(self, compressed, dec):
        """Decode ONE .et stream (uint8 device tensor, 4-byte aligned, holding the file minus its first 4
        bytes -- decode.zig's `compressed_text` -- on every rank) across the ranks of the group, with no
        side information (et_decode_sharded).  The body is cut at multiples of 8 KiB from its 4-byte aligned
        base; every rank synchronises its range (running in from the 16 bytes before it),
        the ranks all-gather (start, exit, symbols), a rank whose start is not its
        predecessor's exit repairs, and when all agree each rank writes its symbols to
        `dec`.  Returns (symbols written by this rank, global index of its first symbol).
        Expected rounds: 1 (a run-in is right 99.6 % of the time on text)."""
        return self.lib_group.decode_sharded(compressed, dec)

    # ------------------------------------------------------------------ concat
    def merge_seams(self, enc, layout):
        """The owner of a word several shards share receives the later shards' bits (in place, in enc)."""
        if not layout["single"]:
            self.lib_group.merge_seams(enc)

    def concat_on_rank0(self, enc, layout):
        """The image as a device tensor on rank 0 (None elsewhere): seam merge + owned words over xGMI (RCCL
        groups) -- what bench.py times as concat_ms.  Without RCCL: through gather_file's host path."""
        if layout["single"]:
            return enc[: layout["et_len"]]
        if self.rccl:
            file_bytes = (layout["starts"][-1] + 7) // 8
            self.merge_seams(enc, layout)
            image = torch.empty((file_bytes + 3) // 4 * 4, dtype=torch.uint8, device=enc.device) if self.rank == 0 else None
            self.lib_group.gather(enc, image, 0)
            return image[:file_bytes] if self.rank == 0 else None
        data = self.gather_file(enc, layout)
        return torch.frombuffer(bytearray(data), dtype=torch.uint8).to(enc.device) if data is not None else None

    def gather_file(self, enc, layout):
        """Bit-offset-adjusted concatenation on rank 0 -> bytes (None elsewhere): seams merged, then
        every rank's owned words into place -- over xGMI (et_shard_gather: RCCL send/recv into an image on
        rank 0's GPU), or, for a process group without RCCL, as one tensor gather over that group (the
        owned word ranges are the library's, et_shard_words)."""
        if layout["single"]:
            return enc[: layout["et_len"]].cpu().numpy().tobytes()
        starts, r = layout["starts"], self.rank
        file_bytes = (starts[-1] + 7) // 8
        self.merge_seams(enc, layout)
        if self.rccl:
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

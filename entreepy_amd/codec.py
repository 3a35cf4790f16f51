"""Host-side mirror of the reference's codec interface over the C ABI.

    reference                                   here
    encode(allocator, text, out_writer,         encode(text, flags) -> bytes
           std_out, flags) !usize  (encode.zig:25)
    decode(allocator, compressed_text, ...)     decode(compressed_text, flags) -> bytes
           !usize                   (decode.zig:13)
    EncodeFlags / DecodeFlags (encode.zig:9-14, decode.zig:7-11)
    error.QueueEmpty on empty input             EmptyInputError

`compressed_text` keeps the reference's convention: the .et file minus its first four
bytes (main.zig:204, test.zig:26).  Everything numeric happens in libentreepy_hip.so;
torch appears only as the owner of device memory and streams in the *_device calls.
"""
import ctypes
from dataclasses import dataclass

import numpy as np

from . import _native as N


class EntreepyError(RuntimeError):
    def __init__(self, status, detail=""):
        msg = N.lib().et_strerror(status).decode()
        super().__init__(f"{msg}{': ' + detail if detail else ''} (et_status {status})")
        self.status = status


class EmptyInputError(EntreepyError):
    """The reference's error.QueueEmpty (queue.zig:28-30 reached from encode.zig:137-138)."""


@dataclass
class EncodeFlags:  # encode.zig:9-14
    write_output: bool = False
    print_output: bool = False
    debug: bool = False


@dataclass
class DecodeFlags:  # decode.zig:7-11
    write_output: bool = False
    print_output: bool = False
    debug: bool = False


def _check(status, ctx=None):
    if status == N.ET_OK:
        return
    detail = N.lib().et_last_error(ctx).decode() if ctx else ""
    if status == N.ET_ERR_EMPTY:
        raise EmptyInputError(status, detail)
    raise EntreepyError(status, detail)


def _host_u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, dtype=np.uint8)
    return a, (a.ctypes.data if a.size else None)


class Codebook:
    """The code table (reference `dictionary`), host side."""

    def __init__(self, raw=None):
        self.raw = raw if raw is not None else N.Codebook()

    @classmethod
    def from_histogram(cls, hist):
        """encode.zig:54-214 via et_build_codebook."""
        h = np.ascontiguousarray(hist, dtype=np.uint64)
        assert h.size == 256
        cb = cls()
        _check(N.lib().et_build_codebook(h.ctypes.data, ctypes.byref(cb.raw)))
        return cb

    @classmethod
    def from_tables(cls, data, length):
        """An arbitrary table (tests of the long-code path)."""
        cb = cls()
        d = np.ascontiguousarray(data, dtype=np.uint32)
        l = np.ascontiguousarray(length, dtype=np.uint8)
        ctypes.memmove(cb.raw.data, d.ctypes.data, 1024)
        ctypes.memmove(cb.raw.length, l.ctypes.data, 256)
        nz = l[l > 0]
        cb.raw.n_coded = int(nz.size)
        cb.raw.min_length = int(nz.min()) if nz.size else 0
        cb.raw.max_length = int(nz.max()) if nz.size else 0
        return cb

    @property
    def data(self):
        return np.frombuffer(self.raw.data, dtype=np.uint32).copy()

    @property
    def length(self):
        return np.frombuffer(self.raw.length, dtype=np.uint8).copy()

    @property
    def dfs_order(self):
        # a lone symbol is a leaf-root with length 0 (encode.zig:137-138): still one dump line
        n = int(self.raw.n_coded) or 1
        return np.frombuffer(self.raw.dfs_order, dtype=np.uint8)[:n].copy()

    def header(self, text_len):
        """encode.zig:259-299 via et_write_header."""
        out = np.zeros(8192, dtype=np.uint8)
        n = ctypes.c_size_t(0)
        _check(N.lib().et_write_header(ctypes.byref(self.raw), int(text_len), out.ctypes.data, out.size, ctypes.byref(n)))
        return out[: n.value].tobytes()

    def bits(self, hist):
        h = np.ascontiguousarray(hist, dtype=np.uint64)
        b = ctypes.c_uint64(0)
        _check(N.lib().et_codebook_bits(ctypes.byref(self.raw), h.ctypes.data, ctypes.byref(b)))
        return b.value


def parse_header(compressed_text):
    """decode.zig:34-141 via et_parse_header -> (Codebook, n_symbols, body_offset)."""
    a, p = _host_u8(compressed_text)
    cb = Codebook()
    n = ctypes.c_uint64(0)
    off = ctypes.c_size_t(0)
    _check(N.lib().et_parse_header(p, a.size, ctypes.byref(cb.raw), ctypes.byref(n), ctypes.byref(off)))
    return cb, n.value, off.value


def encode_bound(n):
    return N.lib().et_encode_bound(n)


class Context:
    """One et_ctx: a GPU's stream, workspaces and pinned staging.

    The *_device calls (and a Group's) take torch tensors, which torch's CURRENT stream produced and will consume, so by
    default they run on that stream: whatever stream is current when the call is made (torch.cuda.stream(...) included) is
    the one the call is ordered on -- no synchronisation is ever needed around them.  (Until round 4 a new context ran on a
    non-blocking stream of its own unless told otherwise, and a caller who forgot use_torch_stream() raced with the kernels
    that made its tensors: tests/soak/soak_sharded.py found that the hard way.)  use_own_stream() / use_stream(ptr) opt out:
    the caller then orders the streams itself.  The C ABI is unchanged: an et_ctx starts on its own stream (et_ctx_set_stream)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        self.device = device
        self._groups = []  # weak references to the Groups made on this context: closed before it
        self._follow_torch = True  # device calls bind the ctx to torch's current stream first
        self._bound = None
        _check(N.lib().et_ctx_create(device, ctypes.byref(self._h)))

    def close(self):
        for ref in self._groups:
            g = ref()
            if g is not None:
                g.close()
        self._groups = []
        if self._h:
            N.lib().et_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- plumbing ---------------------------------------------------------------
    def use_stream(self, hip_stream):
        """Run on a caller-owned stream (a raw hipStream_t); the caller orders it against the streams its tensors live on."""
        self._follow_torch = False
        self._bound = None
        _check(N.lib().et_ctx_set_stream(self._h, ctypes.c_void_p(hip_stream)), self._h)

    def use_own_stream(self):
        """Run on the context's own non-blocking stream; the caller orders it against the streams its tensors live on."""
        self._follow_torch = False
        self._bound = None
        _check(N.lib().et_ctx_use_own_stream(self._h), self._h)

    def use_torch_stream(self):
        """(The default.)  Every device call runs on the stream that is torch's current one when the call is made."""
        self._follow_torch = True
        self._bind()

    def _bind(self):
        """Before a call that takes device tensors: the ctx onto torch's current stream (a pointer compare when nothing changed)."""
        if not self._follow_torch:
            return
        import torch

        ptr = torch.cuda.current_stream(self.device).cuda_stream
        if ptr != self._bound:
            _check(N.lib().et_ctx_set_stream(self._h, ctypes.c_void_p(ptr)), self._h)
            self._bound = ptr

    def reserve(self, max_text_bytes):
        _check(N.lib().et_ctx_reserve(self._h, int(max_text_bytes)), self._h)

    def set_tile_rounds(self, rounds):
        """Force the encode tile to rounds x 4 KiB (0 = size-based).  Test/tuning knob."""
        _check(N.lib().et_ctx_set_tile_rounds(self._h, int(rounds)), self._h)

    TIMING_DECODE_BODY = 2  # et_ctx_enable_timing's ET_TIMING_DECODE_BODY: only the decode's write kernel carries events

    def enable_timing(self, on=True):
        """True / False: every phase / nothing carries HIP events; Context.TIMING_DECODE_BODY: the decode's write kernel alone
        (timings("decode") then holds body_ms, host_ms and the path flags only)."""
        _check(N.lib().et_ctx_enable_timing(self._h, 2 if (on is not True and on == self.TIMING_DECODE_BODY) else int(bool(on))), self._h)

    def timings(self, which=None):
        """Phase timings of the last call (which=None), the last encode-side call ("encode")
        or the last decode ("decode"); waits for that call's last event."""
        t = N.Timings()
        if which is None:
            _check(N.lib().et_last_timings(self._h, ctypes.byref(t)), self._h)
        else:
            _check(N.lib().et_last_timings_of(self._h, {"encode": 0, "decode": 1}[which], ctypes.byref(t)), self._h)
        d = {k: getattr(t, k) for k, _ in N.Timings._fields_ if k not in ("reserved", "pad_")}
        d["exhaustive_sync"] = bool(t.reserved & 1)
        d["tree_walk_sync"] = bool(t.reserved & 2)
        d["chained_write"] = bool(t.reserved & 4)
        d["row_sync"] = bool(t.reserved & 8)
        d["fixed_sync"] = bool(t.reserved & 16)
        d["strips_write"] = bool(t.reserved & 32)
        return d

    def last_codebook(self):
        """Code table of the most recent encode of this context (et_last_codebook)."""
        cb = Codebook()
        _check(N.lib().et_last_codebook(self._h, ctypes.byref(cb.raw)), self._h)
        return cb

    # -- whole calls, files (chunked pinned-buffer pipeline) -------------------------
    def encode_file(self, in_path, out_path=None):
        """c: in_path -> out_path (.et image); out_path None = code only (main.zig -t).
        Returns (bytes read, bytes of .et produced)."""
        return self._file_call(N.lib().et_encode_fd, in_path, out_path, ())

    def decode_file(self, in_path, out_path=None, skip=4):
        """d: in_path (a .et file; its first `skip` bytes are passed over as main.zig:204
        does) -> out_path.  Returns (bytes consumed after the skip, bytes decoded)."""
        return self._file_call(N.lib().et_decode_fd, in_path, out_path, (skip,))

    def _file_call(self, fn, in_path, out_path, extra):
        import os

        fin = os.open(in_path, os.O_RDONLY)
        fout = os.open(out_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644) if out_path is not None else -1
        try:
            a, b = ctypes.c_size_t(0), ctypes.c_size_t(0)
            _check(fn(self._h, fin, *extra, fout, ctypes.byref(a), ctypes.byref(b)), self._h)
            return a.value, b.value
        finally:
            os.close(fin)
            if fout >= 0:
                os.close(fout)

    # -- whole calls, host memory --------------------------------------------------
    def encode(self, text):
        a, p = _host_u8(text)
        out = np.empty(encode_bound(a.size), dtype=np.uint8)
        n = ctypes.c_size_t(0)
        _check(N.lib().et_encode(self._h, p, a.size, out.ctypes.data, out.size, ctypes.byref(n)), self._h)
        return out[: n.value].tobytes()

    def decode(self, compressed_text):
        a, p = _host_u8(compressed_text)
        want = ctypes.c_size_t(0)
        _check(N.lib().et_decoded_size(p, a.size, ctypes.byref(want)), self._h)
        out = np.empty(want.value + 64, dtype=np.uint8)
        n = ctypes.c_size_t(0)
        _check(N.lib().et_decode(self._h, p, a.size, out.ctypes.data, out.size, ctypes.byref(n)), self._h)
        return out[: n.value].tobytes()

    # -- whole calls, device memory (torch uint8 tensors) ------------------------------
    def encode_device(self, text, out):
        """text, out: 1-D uint8 CUDA tensors; out.numel() >= encode_bound(text.numel()).
        Stream-ordered: returns the .et byte count without waiting for the kernels."""
        self._bind()
        n = ctypes.c_size_t(0)
        _check(N.lib().et_encode_device(self._h, text.data_ptr(), text.numel(), out.data_ptr(), out.numel(), ctypes.byref(n)), self._h)
        return n.value

    def decode_device(self, compressed_text, out, skip=0, length=None):
        """compressed_text[skip : skip + length] (default: to its end) -> out; the offsets spare the caller a tensor view."""
        self._bind()
        n = ctypes.c_size_t(0)
        length = compressed_text.numel() - skip if length is None else length
        _check(N.lib().et_decode_device(self._h, compressed_text.data_ptr() + skip, length, out.data_ptr(), out.numel(), ctypes.byref(n)), self._h)
        return n.value

    # -- staged calls (sharded encode) ----------------------------------------------
    def histogram_device(self, text, hist):
        """text: uint8 CUDA tensor; hist: int64/uint64 CUDA tensor of 256 counters."""
        self._bind()
        assert hist.numel() == 256 and hist.element_size() == 8
        _check(N.lib().et_histogram_device(self._h, text.data_ptr(), text.numel(), hist.data_ptr()), self._h)

    def histogram_host(self):
        """The counts of the last histogram_device call on the host (uint64[256]): waits for them, no copy command."""
        c = np.zeros(256, dtype=np.uint64)
        _check(N.lib().et_histogram_host(self._h, c.ctypes.data), self._h)
        return c

    def histogram_on_host(self, counts):
        """The counts of the last histogram_device call, already on the host (uint64[256])."""
        c = np.ascontiguousarray(counts, dtype=np.uint64)
        assert c.size == 256
        _check(N.lib().et_histogram_on_host(self._h, c.ctypes.data), self._h)

    def encode_body_device(self, codebook, text, out, start_bit=0):
        self._bind()
        end = ctypes.c_uint64(0)
        _check(N.lib().et_encode_body_device(self._h, ctypes.byref(codebook.raw), text.data_ptr(), text.numel(), out.data_ptr(),
                                             out.numel() * out.element_size(), int(start_bit), ctypes.byref(end)), self._h)
        return end.value

    def encode_head_shard_device(self, codebook, text, out, header):
        """Shard 0 of a sharded encode: file header followed by the shard's body."""
        self._bind()
        end = ctypes.c_uint64(0)
        hb = np.frombuffer(header, dtype=np.uint8)
        _check(N.lib().et_encode_head_shard_device(self._h, ctypes.byref(codebook.raw), text.data_ptr(), text.numel(), out.data_ptr(),
                                                   out.numel() * out.element_size(), hb.ctypes.data, hb.size, ctypes.byref(end)), self._h)
        return end.value

    def decode_body_device(self, codebook, body, n_symbols, out, start_bit=0):
        self._bind()
        n = ctypes.c_size_t(0)
        _check(N.lib().et_decode_body_device(self._h, ctypes.byref(codebook.raw), body.data_ptr(), body.numel(), int(start_bit),
                                             int(n_symbols), out.data_ptr(), out.numel(), ctypes.byref(n)), self._h)
        return n.value


    def decode_range_sync(self, codebook, stream, begin, end, in_start_bit=-1):
        """One rank's part of a cold multi-GPU decode: synchronise the codewords that begin
        in stream[begin:end] (uint8 device tensor of the body from its 4-byte aligned base;
        begin a multiple of 8192, end too unless it is the stream's end).  Call again with
        the predecessor's exit as in_start_bit to repair.  -> dict(start_bit, exit_bit,
        n_symbols, sweeps)."""
        self._bind()
        info = N.RangeInfo()
        tail = stream.numel() - end
        _check(N.lib().et_decode_range_sync(self._h, ctypes.byref(codebook.raw), stream.data_ptr() + begin, end - begin, tail,
                                            int(begin >= 16), int(in_start_bit), ctypes.byref(info)), self._h)
        return {"start_bit": info.start_bit, "exit_bit": info.exit_bit, "n_symbols": info.n_symbols, "sweeps": info.sweeps,
                "tree_walk": info.reserved == 2}

    def decode_range_maps(self, codebook, stream, begin, end, in_start_bit=-1):
        """Exhaustive variant of decode_range_sync for codes that do not self-synchronise:
        -> (map, n_starts), map[p] = exit bit of the range when its first codeword begins
        p bits in (p < n_starts; constant when in_start_bit >= 0).  Follow with
        decode_range_resolve(start) once the start is known."""
        self._bind()
        m = (ctypes.c_uint8 * 32)()
        k = ctypes.c_uint32(0)
        tail = stream.numel() - end
        _check(N.lib().et_decode_range_maps(self._h, ctypes.byref(codebook.raw), stream.data_ptr() + begin, end - begin, tail, int(in_start_bit),
                                            ctypes.byref(m), ctypes.byref(k)), self._h)
        return bytes(m), k.value

    def decode_range_resolve(self, in_start_bit):
        self._bind()
        info = N.RangeInfo()
        _check(N.lib().et_decode_range_resolve(self._h, int(in_start_bit), ctypes.byref(info)), self._h)
        return {"start_bit": info.start_bit, "exit_bit": info.exit_bit, "n_symbols": info.n_symbols, "sweeps": info.sweeps, "row_walk": info.reserved == 3}

    def decode_range_write(self, max_symbols, out):
        self._bind()
        n = ctypes.c_size_t(0)
        _check(N.lib().et_decode_range_write(self._h, int(max_symbols), out.data_ptr(), out.numel(), ctypes.byref(n)), self._h)
        return n.value


class Group:
    """One rank's et_group: a Context plus how the ranks exchange small host buffers.

    allgather: callable(bytes of this rank) -> bytes of all ranks in rank order (any transport:
    torch.distributed over gloo, MPI, threads ...), or rccl_id: the 128-byte id rank 0 got from
    Group.rccl_unique_id(), for an RCCL communicator of the library's own over xGMI.

    All ranks make the same calls in the same order.  A call in which any rank fails raises on EVERY rank
    (the rows of each exchange carry the ranks' statuses; include/entreepy_hip.h, "FAILURES").

    lib: the loaded library whose group entry points are called (default: libentreepy_hip.so).  The CPU tests
    pass a build of the same sequence (csrc/et_shard_seq.cpp) over a stand-in for the GPU."""

    def __init__(self, ctx, rank, world, allgather=None, rccl_id=None, lib=None):
        import weakref

        self.ctx, self.rank, self.world = ctx, rank, world
        self._lib = lib if lib is not None else N.lib()
        self._h = ctypes.c_void_p()
        self._py_gather = allgather
        self._cb = N.ALLGATHER_FN(self._gather)  # kept alive with the group
        ctx_h = getattr(ctx, "_h", None)
        if rccl_id is not None:
            ident = (ctypes.c_uint8 * N.ET_RCCL_ID_BYTES).from_buffer_copy(bytes(rccl_id))
            _check(self._lib.et_group_create_rccl(ctx_h, rank, world, ctypes.byref(ident), ctypes.byref(self._h)), ctx_h)
        else:
            _check(self._lib.et_group_create(ctx_h, rank, world, self._cb if allgather else ctypes.cast(None, N.ALLGATHER_FN), None, ctypes.byref(self._h)),
                   ctx_h if lib is None else None)
        if hasattr(ctx, "_groups"):
            ctx._groups.append(weakref.ref(self))

    @staticmethod
    def rccl_unique_id():
        ident = (ctypes.c_uint8 * N.ET_RCCL_ID_BYTES)()
        _check(N.lib().et_rccl_unique_id(ctypes.byref(ident)))
        return bytes(ident)

    def _gather(self, user, send, recv, nbytes):
        try:
            got = self._py_gather(ctypes.string_at(send, nbytes))
            if len(got) != nbytes * self.world:
                return 1
            ctypes.memmove(recv, got, len(got))
            return 0
        except Exception:  # noqa: BLE001 -- reported to the C caller as a failed exchange
            return 1

    def _bind_ctx(self):
        """The group's context onto torch's current stream (Context._bind); the CPU stand-in's contexts have no stream."""
        b = getattr(self.ctx, "_bind", None)
        if b is not None:
            b()

    def _ck(self, status):
        if status == N.ET_OK:
            return
        detail = self._lib.et_group_last_error(self._h).decode()
        if status == N.ET_ERR_EMPTY:
            raise EmptyInputError(status, detail)
        raise EntreepyError(status, detail)

    def close(self):
        if self._h:
            self._lib.et_group_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def force_collectives(self, on=True):
        """A group of one takes the transport's path all the same (one-GPU tests of the N > 1 path)."""
        self._ck(self._lib.et_group_set_option(self._h, N.ET_GROUP_FORCE_COLLECTIVES, int(bool(on))))

    def set_timeout_ms(self, ms):
        self._ck(self._lib.et_group_set_option(self._h, N.ET_GROUP_TIMEOUT_MS, int(ms)))

    @staticmethod
    def _info(i):
        return {k: getattr(i, k) for k, _ in N.ShardInfo._fields_}

    @staticmethod
    def _ptr(t):
        return t.data_ptr() if t is not None and t.numel() else None

    def encode_sharded(self, text, out):
        """et_encode_sharded: this rank's chunk (uint8 tensor, may be empty) -> its piece of the image in `out`."""
        self._bind_ctx()
        i = N.ShardInfo()
        self._ck(self._lib.et_encode_sharded(self._h, self._ptr(text), text.numel(), out.data_ptr() if out is not None else None,
                                             out.numel() if out is not None else 0, ctypes.byref(i)))
        return self._info(i)

    def merge_seams(self, out):
        self._bind_ctx()
        self._ck(self._lib.et_shard_merge_seams(self._h, out.data_ptr() if out is not None else None))

    def write_fd(self, out, fd):
        self._bind_ctx()
        self._ck(self._lib.et_shard_write_fd(self._h, out.data_ptr(), fd))

    def place(self, out, image):
        self._bind_ctx()
        self._ck(self._lib.et_shard_place(self._h, out.data_ptr(), image.data_ptr(), image.numel()))

    def gather(self, out, image, root=0):
        self._bind_ctx()
        self._ck(self._lib.et_shard_gather(self._h, out.data_ptr(), image.data_ptr() if image is not None else None,
                                           image.numel() if image is not None else 0, root))

    def info(self):
        i = N.ShardInfo()
        self._ck(self._lib.et_group_last_info(self._h, ctypes.byref(i)))
        return self._info(i)

    def codebook(self):
        cb = Codebook()
        self._ck(self._lib.et_group_codebook(self._h, ctypes.byref(cb.raw)))
        return cb

    def start_bits(self):
        a = np.zeros(self.world + 1, dtype=np.uint64)
        self._ck(self._lib.et_group_start_bits(self._h, a.ctypes.data))
        return [int(x) for x in a]

    def decode_sharded(self, compressed_text, out):
        """et_decode_sharded: this rank's block range of one cold stream -> (symbols written, index of the first)."""
        self._bind_ctx()
        n = ctypes.c_size_t(0)
        first = ctypes.c_uint64(0)
        self._ck(self._lib.et_decode_sharded(self._h, self._ptr(compressed_text), compressed_text.numel() if compressed_text is not None else 0,
                                             out.data_ptr() if out is not None else None, out.numel() if out is not None else 0, ctypes.byref(n), ctypes.byref(first)))
        return n.value, first.value

    def decode_window(self, head, length):
        """et_decode_shard_window: (offset, bytes) of `compressed` this rank needs besides the dictionary (head =
        the first min(length, 8192) bytes of it, host bytes)."""
        a, p = _host_u8(head)
        off, ln = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._ck(self._lib.et_decode_shard_window(p, a.size, int(length), self.rank, self.world, ctypes.byref(off), ctypes.byref(ln)))
        return off.value, ln.value

    def decode_begin(self, head, length, window, window_off, cap=None):
        """et_decode_sharded_begin (collective): window = uint8 tensor holding bytes [window_off, ...) of `compressed`
        -> (symbols this rank writes, index of the first)."""
        self._bind_ctx()
        a, p = _host_u8(head) if head is not None else (np.zeros(0, dtype=np.uint8), None)
        n, first = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._ck(self._lib.et_decode_sharded_begin(self._h, p, a.size, int(length), self._ptr(window), int(window_off), window.numel() if window is not None else 0,
                                                   (1 << 64) - 1 if cap is None else int(cap), ctypes.byref(n), ctypes.byref(first)))
        return n.value, first.value

    def decode_write(self, out):
        self._bind_ctx()
        n = ctypes.c_size_t(0)
        self._ck(self._lib.et_decode_sharded_write(self._h, out.data_ptr() if out is not None else None, out.numel() if out is not None else 0, ctypes.byref(n)))
        return n.value


def shard_words(starts, world, rank):
    """et_shard_words: (piece_lo, piece_hi, owned_lo, owned_hi) of `rank`, in 4-byte words of the image."""
    a = np.ascontiguousarray(starts, dtype=np.uint64)
    w = np.zeros(4, dtype=np.uint64)
    _check(N.lib().et_shard_words(a.ctypes.data, world, rank, w.ctypes.data))
    return tuple(int(x) for x in w)


def seam_word(starts, world, rank, first_last):
    """et_seam_word: the word closing `rank`'s owned range merged with the later shards that begin in it, or None."""
    a = np.ascontiguousarray(starts, dtype=np.uint64)
    fl = np.ascontiguousarray(first_last, dtype=np.uint32)
    merged, has = ctypes.c_uint32(0), ctypes.c_int(0)
    _check(N.lib().et_seam_word(a.ctypes.data, world, rank, fl.ctypes.data, ctypes.byref(merged), ctypes.byref(has)))
    return merged.value if has.value else None


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def encode(text, flags=None, device=0):
    """encode.zig:25.  Returns the .et file image (the bytes the reference hands to
    out_writer.writeAll, encode.zig:319).  Raises EmptyInputError on b''."""
    return default_context(device).encode(text)


def decode(compressed_text, flags=None, device=0):
    """decode.zig:13.  `compressed_text` = .et file minus its first 4 bytes."""
    return default_context(device).decode(compressed_text)

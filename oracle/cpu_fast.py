"""The fast chunk-parallel CPU variant (oracle/et_cpu_fast.c), one chunk per host thread --
MEASUREMENT INFRASTRUCTURE ONLY: bench.py's all-cores CPU baseline and the
tests that pin it to the restatement.  ctypes releases the GIL, so plain Python threads run
the chunks in parallel; the steps in between (code table, offsets, merging the chunk
starts) are the cheap sequential ones.
"""
import ctypes
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import oracle as O

MARKS = 1 << 15  # bits at the front of each chunk in which the true start may meet the first walk
NO_MARK = 0xFFFFFFFF


def _lib():
    L = O.lib()
    if not hasattr(L, "_fast_ready"):
        vp, u64, u32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32
        L.et_oracle_parse_dict.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(O._Dict), ctypes.POINTER(u64), ctypes.POINTER(u32)]
        L.et_oracle_parse_dict.restype = ctypes.c_int64
        L.et_fast_tables_size.restype = ctypes.c_size_t
        L.et_fast_build_tables.argtypes = [ctypes.POINTER(O._Dict), vp]
        L.et_fast_build_tables.restype = ctypes.c_int
        L.et_fast_pack.argtypes = [ctypes.POINTER(O._Dict), vp, ctypes.c_size_t, vp, u64]
        L.et_fast_pack.restype = u64
        L.et_fast_walk.argtypes = [vp, vp, u64, u64, u64, u64, vp, ctypes.POINTER(u64), vp, u64, u32]
        L.et_fast_walk.restype = u64
        L.et_fast_merge.argtypes = [vp, vp, u64, u64, vp, u64, u32, ctypes.POINTER(u64), ctypes.POINTER(u32)]
        L.et_fast_merge.restype = ctypes.c_int
        L._fast_ready = True
    return L


def _cuts(n, parts):
    return [n * k // parts for k in range(parts + 1)]


def encode(text, threads):
    """-> .et bytes, identical to oracle.encode(text)."""
    L = _lib()
    a = O._as_u8(text)
    cuts = _cuts(a.size, max(1, min(threads, a.size // 4096 or 1)))
    parts = len(cuts) - 1
    with ThreadPoolExecutor(parts) as pool:
        hists = list(pool.map(lambda k: O.histogram(a[cuts[k]:cuts[k + 1]]), range(parts)))
        data, length, _ = O.build_dict(np.sum(hists, axis=0, dtype=np.uint64))
        header = O.write_header(data, length, a.size)
        d = O._mk_dict(data, length)
        bits = [int((h * length.astype(np.uint64)).sum()) for h in hists]
        starts = [8 * len(header)]
        for b in bits:
            starts.append(starts[-1] + b)
        out = np.empty((starts[-1] + 7) // 8 + 8, dtype=np.uint8)
        out[: len(header)] = np.frombuffer(header, dtype=np.uint8)
        for s in starts:  # bytes two chunks may share are OR-ed into
            out[s >> 3] = 0
        ends = list(pool.map(lambda k: L.et_fast_pack(ctypes.byref(d), a[cuts[k]:].ctypes.data, cuts[k + 1] - cuts[k],
                                                      out.ctypes.data, starts[k]), range(parts)))
    assert ends == starts[1:]
    return out[: (starts[-1] + 7) // 8].tobytes()  # encode.zig:317-319: flushed (zero pad, Q13), then bits_written / 8 bytes


def decode(compressed, threads):
    """`compressed` = file[4:] -> the text, identical to oracle.decode(compressed) on well-formed streams."""
    L = _lib()
    ct = O._as_u8(compressed)
    d = O._Dict()
    body_start, body_len = ctypes.c_uint64(0), ctypes.c_uint32(0)
    if L.et_oracle_parse_dict(ct.ctypes.data, ct.size, ctypes.byref(d), ctypes.byref(body_start), ctypes.byref(body_len)):
        raise O.OracleError(O.FORMAT)
    tables = np.zeros(L.et_fast_tables_size(), dtype=np.uint8)
    if L.et_fast_build_tables(ctypes.byref(d), tables.ctypes.data):
        raise O.OracleError(O.FORMAT)
    body = ct[body_start.value:]
    n_sym = body_len.value
    total_bits = body.size * 8
    if n_sym == 0 or total_bits == 0:
        return b""
    parts = max(1, min(threads, body.size // 65536 or 1))
    cut = [8 * c for c in _cuts(body.size, parts)]  # chunk boundaries in bits, on bytes
    tp, bp = tables.ctypes.data, body.ctypes.data
    big = 1 << 62

    def first_walk(k):
        marks = np.full(MARKS, NO_MARK, dtype=np.uint32)
        ex = ctypes.c_uint64(0)
        c = L.et_fast_walk(tp, bp, body.size, cut[k], cut[k + 1], big, None, ctypes.byref(ex), marks.ctypes.data, cut[k], MARKS)
        return c, ex.value, marks

    with ThreadPoolExecutor(parts) as pool:
        walks = list(pool.map(first_walk, range(parts)))
        # the true start of chunk k is the true exit of chunk k-1
        start, count = [0] * parts, [0] * parts
        count[0], exit_bit, _ = walks[0]
        for k in range(1, parts):
            c, ex, marks = walks[k]
            start[k] = exit_bit
            extra, at = ctypes.c_uint64(0), ctypes.c_uint32(0)
            if exit_bit >= cut[k + 1]:  # a code spans the whole chunk (tiny chunks only)
                count[k] = 0
            elif L.et_fast_merge(tp, bp, body.size, exit_bit, marks.ctypes.data, cut[k], MARKS, ctypes.byref(extra), ctypes.byref(at)):
                count[k] = extra.value + c - int(marks[at.value])
                exit_bit = ex
            else:  # no meeting point in the window: walk the chunk again from its true start
                ex2 = ctypes.c_uint64(0)
                count[k] = L.et_fast_walk(tp, bp, body.size, exit_bit, cut[k + 1], big, None, ctypes.byref(ex2), None, 0, 0)
                exit_bit = ex2.value
        offs = [0]
        for c in count:
            offs.append(offs[-1] + c)
        n_out = min(n_sym, offs[-1])
        out = np.empty(n_out + 8, dtype=np.uint8)

        def write(k):
            lim = max(0, min(count[k], n_out - offs[k]))
            if lim:
                ex = ctypes.c_uint64(0)
                got = L.et_fast_walk(tp, bp, body.size, start[k], big, lim, out[offs[k]:].ctypes.data, ctypes.byref(ex), None, 0, 0)
                assert got == lim
        list(pool.map(write, range(parts)))
    return out[:n_out].tobytes()

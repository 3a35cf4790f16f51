"""ctypes wrapper over oracle/libet_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (see oracle/et_oracle.h).  The product path lives in entreepy_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libet_oracle.so")

OK, QUEUE_EMPTY, NO_SPACE, HANG, OOB, FORMAT = 0, 1, 2, 3, 4, 5


class OracleError(Exception):
    def __init__(self, status):
        super().__init__({1: "QueueEmpty", 2: "NoSpaceLeft", 3: "Hang", 4: "OutOfBounds", 5: "Format"}.get(status, str(status)))
        self.status = status


def build():
    """Compile the oracle with gcc (no-op when the .so is newer than its sources)."""
    srcs = [os.path.join(_HERE, f) for f in ("et_oracle.c", "et_cpu_fast.c", "et_oracle.h")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "libet_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class _Dict(ctypes.Structure):
    _fields_ = [("data", ctypes.c_uint32 * 256), ("length", ctypes.c_uint8 * 256)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p = ctypes.c_void_p
        L.et_oracle_histogram.argtypes = [u8p, ctypes.c_size_t, ctypes.c_void_p]
        L.et_oracle_histogram.restype = None
        L.et_oracle_build_dict.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Dict), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.et_oracle_build_dict.restype = ctypes.c_int
        L.et_oracle_write_header.argtypes = [ctypes.POINTER(_Dict), ctypes.c_uint64, u8p, ctypes.c_size_t]
        L.et_oracle_write_header.restype = ctypes.c_int64
        L.et_oracle_pack_body.argtypes = [ctypes.POINTER(_Dict), u8p, ctypes.c_size_t, u8p, ctypes.c_size_t, ctypes.c_uint64]
        L.et_oracle_pack_body.restype = ctypes.c_int64
        for name in ("et_oracle_encode", "et_oracle_decode_ref", "et_oracle_decode"):
            f = getattr(L, name)
            f.argtypes = [u8p, ctypes.c_size_t, u8p, ctypes.c_size_t]
            f.restype = ctypes.c_int64
        L.et_oracle_format_file_size.argtypes = [ctypes.c_float, ctypes.c_char_p, ctypes.c_size_t]
        L.et_oracle_format_file_size.restype = None
        _lib = L
    return _lib


def _as_u8(x):
    if isinstance(x, np.ndarray):
        a = np.ascontiguousarray(x, dtype=np.uint8)
    else:
        a = np.frombuffer(bytes(x), dtype=np.uint8)
    return a


def _ptr(a):
    return a.ctypes.data if a.size else None


def histogram(text):
    a = _as_u8(text)
    occ = np.zeros(256, dtype=np.uint64)
    lib().et_oracle_histogram(_ptr(a), a.size, occ.ctypes.data)
    return occ


def build_dict(occ):
    """-> (data[256] u32, length[256] u8, dfs_order bytes).  Raises OracleError(QUEUE_EMPTY)."""
    occ = np.ascontiguousarray(occ, dtype=np.uint64)
    d = _Dict()
    order = np.zeros(256, dtype=np.uint8)
    nl = ctypes.c_int(0)
    rc = lib().et_oracle_build_dict(occ.ctypes.data, ctypes.byref(d), order.ctypes.data, ctypes.byref(nl))
    if rc:
        raise OracleError(rc)
    return (np.frombuffer(d.data, dtype=np.uint32).copy(), np.frombuffer(d.length, dtype=np.uint8).copy(),
            order[: nl.value].copy())


def _mk_dict(data, length):
    d = _Dict()
    ctypes.memmove(d.data, np.ascontiguousarray(data, dtype=np.uint32).ctypes.data, 1024)
    ctypes.memmove(d.length, np.ascontiguousarray(length, dtype=np.uint8).ctypes.data, 256)
    return d


def write_header(data, length, text_len):
    d = _mk_dict(data, length)
    out = np.zeros(8192, dtype=np.uint8)
    r = lib().et_oracle_write_header(ctypes.byref(d), int(text_len), out.ctypes.data, out.size)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def pack_body(data, length, text, start_bit=0, cap=None):
    """Pack `text` with an arbitrary code table into a zeroed buffer at start_bit.
    -> (bytes up to the last touched byte, end_bit)."""
    a = _as_u8(text)
    d = _mk_dict(data, length)
    if cap is None:
        cap = (start_bit + int(np.asarray(length, dtype=np.uint64)[a].sum()) + 7) // 8 + 8
    out = np.zeros(cap, dtype=np.uint8)
    r = lib().et_oracle_pack_body(ctypes.byref(d), _ptr(a), a.size, out.ctypes.data, out.size, int(start_bit))
    if r < 0:
        raise OracleError(-r)
    return out[: (r + 7) // 8].tobytes(), r


def encode(text):
    """encode.zig:25 whole call -> .et bytes.  Raises OracleError(QUEUE_EMPTY) on empty input."""
    a = _as_u8(text)
    out = np.empty(a.size + 7200, dtype=np.uint8)  # encode.zig:253-254
    r = lib().et_oracle_encode(_ptr(a), a.size, out.ctypes.data, out.size)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def _decode(fn, compressed, cap):
    a = _as_u8(compressed)
    if cap is None:
        cap = int.from_bytes(a[1:5].tobytes(), "big") + 64 if a.size >= 5 else 64
    out = np.empty(max(cap, 1), dtype=np.uint8)
    r = fn(_ptr(a), a.size, out.ctypes.data, cap)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def decode_ref(compressed, cap=None):
    """LITERAL decode.zig restatement (quirks included).  `compressed` = file[4:]."""
    return _decode(lib().et_oracle_decode_ref, compressed, cap)


def decode(compressed, cap=None):
    """Intended inverse of encode().  `compressed` = file[4:]."""
    return _decode(lib().et_oracle_decode, compressed, cap)


def format_file_size(nbytes):
    buf = ctypes.create_string_buffer(64)
    lib().et_oracle_format_file_size(float(nbytes), buf, 64)
    return buf.value.decode()

"""ctypes binding of libentreepy_hip.so (include/entreepy_hip.h).

The library is the product; this module only declares its C ABI.  There is no
Python or CPU fallback: if the shared object is missing, or a call needs a GPU that
is not there, the error is raised to the caller.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ET_LIB_PATH: load another build of the same library (A/B variants in tools/ab_variants.sh)
LIB_PATH = os.environ.get("ET_LIB_PATH") or os.path.join(_HERE, "libentreepy_hip.so")

ET_OK, ET_ERR_EMPTY, ET_ERR_NOMEM, ET_ERR_CAP, ET_ERR_FORMAT, ET_ERR_HIP, ET_ERR_ARG, ET_ERR_UNSUPPORTED, ET_ERR_IO = range(9)


class Codebook(ctypes.Structure):
    """struct et_codebook: the reference's dictionary[256] of Code (encode.zig:141-146)."""

    _fields_ = [
        ("data", ctypes.c_uint32 * 256),
        ("length", ctypes.c_uint8 * 256),
        ("dfs_order", ctypes.c_uint8 * 256),
        ("n_coded", ctypes.c_uint32),
        ("min_length", ctypes.c_uint32),
        ("max_length", ctypes.c_uint32),
    ]


class Timings(ctypes.Structure):
    _fields_ = [
        ("hist_ms", ctypes.c_float),
        ("host_ms", ctypes.c_float),
        ("scan_ms", ctypes.c_float),
        ("body_ms", ctypes.c_float),
        ("sync_ms", ctypes.c_float),
        ("total_ms", ctypes.c_float),
        ("sync_iters", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("sync_first_ms", ctypes.c_float),
        ("pad_", ctypes.c_uint32),
    ]


class RangeInfo(ctypes.Structure):
    _fields_ = [
        ("start_bit", ctypes.c_uint32),
        ("exit_bit", ctypes.c_uint32),
        ("n_symbols", ctypes.c_uint64),
        ("sweeps", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
    ]


_vp, _sz, _u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64
_szp = ctypes.POINTER(ctypes.c_size_t)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_cbp = ctypes.POINTER(Codebook)

# name -> (restype, argtypes); every symbol include/entreepy_hip.h declares.
SIGNATURES = {
    "et_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "et_ctx_destroy": (None, [_vp]),
    "et_ctx_set_stream": (ctypes.c_int, [_vp, _vp]),
    "et_ctx_use_own_stream": (ctypes.c_int, [_vp]),
    "et_ctx_reserve": (ctypes.c_int, [_vp, _sz]),
    "et_ctx_set_tile_rounds": (ctypes.c_int, [_vp, ctypes.c_uint32]),
    "et_ctx_enable_timing": (ctypes.c_int, [_vp, ctypes.c_int]),
    "et_last_timings": (ctypes.c_int, [_vp, ctypes.POINTER(Timings)]),
    "et_last_timings_of": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(Timings)]),
    "et_last_codebook": (ctypes.c_int, [_vp, _cbp]),
    "et_last_error": (ctypes.c_char_p, [_vp]),
    "et_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "et_version": (ctypes.c_char_p, []),
    "et_check_magic": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p)]),
    "et_selftest_decode_tables": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Codebook), ctypes.POINTER(ctypes.c_int)]),
    "et_encode_bound": (_sz, [_sz]),
    "et_encode_fd": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "et_decode_fd": (ctypes.c_int, [_vp, ctypes.c_int, _sz, ctypes.c_int, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "et_encode": (ctypes.c_int, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "et_decode": (ctypes.c_int, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "et_decoded_size": (ctypes.c_int, [_vp, _sz, _szp]),
    "et_encode_device": (ctypes.c_int, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "et_decode_device": (ctypes.c_int, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "et_histogram_device": (ctypes.c_int, [_vp, _vp, _sz, _vp]),
    "et_histogram_on_host": (ctypes.c_int, [_vp, _vp]),
    "et_build_codebook": (ctypes.c_int, [_vp, _cbp]),
    "et_write_header": (ctypes.c_int, [_cbp, _u64, _vp, _sz, _szp]),
    "et_codebook_bits": (ctypes.c_int, [_cbp, _vp, _u64p]),
    "et_plan_shards": (ctypes.c_int, [_vp, ctypes.c_uint32, _cbp, _vp, _sz, ctypes.POINTER(ctypes.c_size_t), _vp]),
    "et_encode_body_device": (ctypes.c_int, [_vp, _cbp, _vp, _sz, _vp, _sz, _u64, _u64p]),
    "et_encode_head_shard_device": (ctypes.c_int, [_vp, _cbp, _vp, _sz, _vp, _sz, _vp, _sz, _u64p]),
    "et_parse_header": (ctypes.c_int, [_vp, _sz, _cbp, _u64p, _szp]),
    "et_decode_range_sync": (ctypes.c_int, [_vp, _cbp, _vp, _sz, _sz, ctypes.c_int, ctypes.c_int32, ctypes.POINTER(RangeInfo)]),
    "et_decode_range_maps": (ctypes.c_int, [_vp, _cbp, _vp, _sz, _sz, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint8 * 32), ctypes.POINTER(ctypes.c_uint32)]),
    "et_decode_range_resolve": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.POINTER(RangeInfo)]),
    "et_decode_range_write": (ctypes.c_int, [_vp, _u64, _vp, _sz, _szp]),
    "et_decode_body_device": (ctypes.c_int, [_vp, _cbp, _vp, _sz, ctypes.c_uint32, _u64, _vp, _sz, _szp]),
}

_lib = None


def lib():
    """Load libentreepy_hip.so.  torch is imported first so that the library binds to
    the HIP runtime already in the process (same libamdhip64.so.7 soname) and shares
    streams and device memory with it."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C entreepy_amd/csrc`).  entreepy_amd has no fallback path."
            )
        try:
            import torch  # noqa: F401  (loads the process-wide HIP runtime)
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib

"""ctypes binding of libentreepy_hip.so (include/entreepy_hip.h).

The library is the product; this module only declares its C ABI.  There is no
Python or CPU fallback: if the shared object is missing, or a call needs a GPU that
is not there, the error is raised to the caller.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ET_LIB_PATH: load another build of the same library (A/B runs of two builds in one tree)
LIB_PATH = os.environ.get("ET_LIB_PATH") or os.path.join(_HERE, "libentreepy_hip.so")

ET_OK, ET_ERR_EMPTY, ET_ERR_NOMEM, ET_ERR_CAP, ET_ERR_FORMAT, ET_ERR_HIP, ET_ERR_ARG, ET_ERR_UNSUPPORTED, ET_ERR_IO, ET_ERR_RCCL = range(10)
ET_RCCL_ID_BYTES = 128
ET_PATH_TREE_WALK, ET_PATH_EXIT_MAPS, ET_PATH_ROWS, ET_PATH_FIXED, ET_PATH_WINDOWS = range(5)  # et_decode_path


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


class ShardInfo(ctypes.Structure):
    """struct et_shard_info: where a rank's shard sits in the .et image."""

    _fields_ = [
        ("start_bit", ctypes.c_uint64),
        ("end_bit", ctypes.c_uint64),
        ("local_start_bit", ctypes.c_uint64),
        ("header_len", ctypes.c_uint64),
        ("file_bytes", ctypes.c_uint64),
        ("text_len", ctypes.c_uint64),
        ("piece_word_lo", ctypes.c_uint64),
        ("piece_word_hi", ctypes.c_uint64),
        ("owned_word_lo", ctypes.c_uint64),
        ("owned_word_hi", ctypes.c_uint64),
        ("exchange_ms", ctypes.c_float),
        ("plan_ms", ctypes.c_float),
        ("seam_ms", ctypes.c_float),
        ("concat_ms", ctypes.c_float),
    ]


# int (*et_allgather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

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
    "et_prefix_collisions": (ctypes.c_int, [_cbp, _vp, _sz, _szp]),
    "et_check_magic": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p)]),
    "et_selftest_decode_tables": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Codebook), ctypes.POINTER(ctypes.c_int)]),
    "et_treewalk_table": (ctypes.c_int, [_cbp, _vp, _sz, ctypes.POINTER(ctypes.c_uint32)]),
    "et_selftest_treewalk_table": (ctypes.c_int, [_vp, _cbp, ctypes.POINTER(ctypes.c_uint32)]),
    "et_chain_tables": (ctypes.c_int, [_cbp, _vp, _sz, ctypes.POINTER(ctypes.c_uint32), _vp, _vp, _sz, ctypes.POINTER(ctypes.c_uint32)]),
    "et_row_code": (ctypes.c_int, [_cbp, ctypes.POINTER(ctypes.c_uint32)]),
    "et_decode_path": (ctypes.c_int, [_cbp, ctypes.POINTER(ctypes.c_uint32)]),
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
    "et_histogram_host": (ctypes.c_int, [_vp, _vp]),
    "et_histogram_device_ptr": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_void_p)]),
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
    "et_ctx_stream": (_vp, [_vp]),
    "et_ctx_device": (ctypes.c_int, [_vp]),
    "et_device_to_fd": (ctypes.c_int, [_vp, _vp, _sz, ctypes.c_int, _u64]),
    "et_fd_to_device": (ctypes.c_int, [_vp, ctypes.c_int, _u64, _sz, _vp]),
    "et_group_create": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ALLGATHER_FN, _vp, ctypes.POINTER(_vp)]),
    "et_rccl_unique_id": (ctypes.c_int, [_vp]),
    "et_group_create_rccl": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.POINTER(_vp)]),
    "et_group_destroy": (None, [_vp]),
    "et_group_last_error": (ctypes.c_char_p, [_vp]),
    "et_encode_sharded": (ctypes.c_int, [_vp, _vp, _sz, _vp, _sz, ctypes.POINTER(ShardInfo)]),
    "et_shard_merge_seams": (ctypes.c_int, [_vp, _vp]),
    "et_shard_write_fd": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    "et_shard_place": (ctypes.c_int, [_vp, _vp, _vp, _sz]),
    "et_shard_gather": (ctypes.c_int, [_vp, _vp, _vp, _sz, ctypes.c_int]),
    "et_group_codebook": (ctypes.c_int, [_vp, _cbp]),
    "et_group_start_bits": (ctypes.c_int, [_vp, _vp]),
    "et_group_last_info": (ctypes.c_int, [_vp, ctypes.POINTER(ShardInfo)]),
    "et_shard_words": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, _vp]),
    "et_seam_word": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, _vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int)]),
    "et_decode_sharded": (ctypes.c_int, [_vp, _vp, _sz, _vp, _sz, _szp, _u64p]),
    "et_decode_shard_window": (ctypes.c_int, [_vp, _sz, _u64, ctypes.c_int, ctypes.c_int, _u64p, _u64p]),
    "et_decode_sharded_begin": (ctypes.c_int, [_vp, _vp, _sz, _u64, _vp, _u64, _sz, _u64, _u64p, _u64p]),
    "et_decode_sharded_write": (ctypes.c_int, [_vp, _vp, _sz, _szp]),
    "et_group_set_option": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64]),
    "et_rccl_library": (ctypes.c_char_p, []),
}

ET_GROUP_FORCE_COLLECTIVES, ET_GROUP_TIMEOUT_MS = 1, 2
# The entry points of a group (csrc/et_shard_seq.cpp + a backend): what a stand-in library of the CPU tests also exports.
GROUP_SYMBOLS = ("et_group_create", "et_group_destroy", "et_group_last_error", "et_group_set_option", "et_encode_sharded", "et_shard_merge_seams", "et_shard_write_fd",
                 "et_shard_place", "et_shard_gather", "et_group_codebook", "et_group_start_bits", "et_group_last_info", "et_decode_sharded", "et_decode_shard_window",
                 "et_decode_sharded_begin", "et_decode_sharded_write")


def declare(L, names=None):
    """restype / argtypes of the named entry points (default: all) on a loaded library."""
    for name in names or SIGNATURES:
        res, args = SIGNATURES[name]
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


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
        _lib = declare(L)
    return _lib

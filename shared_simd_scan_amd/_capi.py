"""ctypes binding of include/mi355_scan.h.  No compute happens in Python; a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

_lib = None

OK = 0
LAYOUT_PER_PREDICATE = 0
LAYOUT_LINEAR = 1
GEN_MOD, GEN_SPLITMIX, GEN_INDEX = 0, 1, 2
COMM_ID_BYTES = 128

# every symbol include/mi355_scan.h declares: (name, restype, argtypes)
_vp, _u64, _u32, _i32, _sz, _int = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_size_t, C.c_int
SYMBOLS = [
    ("mi355_last_error", C.c_char_p, []),
    ("mi355_version", C.c_char_p, []),
    ("mi355_ctx_create", _int, [_int, _vp, C.POINTER(_vp)]),
    ("mi355_ctx_destroy", _int, [_vp]),
    ("mi355_ctx_synchronize", _int, [_vp]),
    ("mi355_device_count", _int, [C.POINTER(_int)]),
    ("mi355_ctx_set_option", _int, [_vp, C.c_char_p, _int]),
    ("mi355_ctx_set_stream", _int, [_vp, _vp]),
    ("mi355_tune_dev", _int, [_vp, _vp, _u64, C.c_uint, C.c_uint]),
    ("mi355_tuned_blocks_per_cu", _int, [_vp, C.c_uint, C.c_uint, _int]),
    ("mi355_shard_rows", _int, [_u64, C.c_uint, C.c_uint, C.POINTER(_u64), C.POINTER(_u64)]),
    ("mi355_compressed_buffer_size", _sz, [C.c_uint, _sz]),
    ("mi355_decompression_output_buffer_size", _sz, [_sz]),
    ("mi355_scan_output_buffer_size", _sz, [_sz]),
    ("mi355_bitmap_stride", _sz, [_sz]),
    ("mi355_dev_alloc", _int, [_vp, _sz, C.POINTER(_vp)]),
    ("mi355_dev_free", _int, [_vp, _vp]),
    ("mi355_dev_upload", _int, [_vp, _vp, _vp, _sz]),
    ("mi355_dev_download", _int, [_vp, _vp, _vp, _sz]),
    ("mi355_dev_memset", _int, [_vp, _vp, _int, _sz]),
    ("mi355_pack_u16", _int, [_vp, _vp, _u64, C.c_uint, _vp]),
    ("mi355_pack_u32", _int, [_vp, _vp, _u64, C.c_uint, _vp]),
    ("mi355_pack_u16_dev", _int, [_vp, _vp, _u64, C.c_uint, _vp]),
    ("mi355_pack_u32_dev", _int, [_vp, _vp, _u64, C.c_uint, _vp]),
    ("mi355_generate_dev", _int, [_vp, _int, _u64, _u64, C.c_uint, _u64, _vp]),
    ("mi355_decompress", _int, [_vp, _vp, _u64, C.c_uint, _vp]),
    ("mi355_decompress_dev", _int, [_vp, _vp, _u64, C.c_uint, _vp]),
    ("mi355_scan_eq", _int, [_vp, _vp, _u64, C.c_uint, _i32, _vp, C.POINTER(_u64)]),
    ("mi355_scan_eq_dev", _int, [_vp, _vp, _u64, C.c_uint, _i32, _vp, _vp]),
    ("mi355_scan_range", _int, [_vp, _vp, _u64, C.c_uint, _u32, _u32, _vp, C.POINTER(_u64)]),
    ("mi355_scan_range_dev", _int, [_vp, _vp, _u64, C.c_uint, _u32, _u32, _vp, _vp]),
    ("mi355_shared_scan_eq", _int, [_vp, _vp, _u64, C.c_uint, _vp, C.c_uint, _vp, _vp]),
    ("mi355_shared_scan_eq_linear", _int, [_vp, _vp, _u64, C.c_uint, _vp, C.c_uint, _vp, _vp]),
    ("mi355_shared_scan_eq_dev", _int, [_vp, _vp, _u64, C.c_uint, _vp, C.c_uint, _int, _vp, _u64, _vp]),
    ("mi355_scan_where_dev", _int, [_vp, _vp, _u64, C.c_uint, _int, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    ("mi355_scan_combine_dev", _int, [_vp, _vp, _u64, C.c_uint, _int, C.c_int64, C.c_int64, _int, _vp, _vp, _vp]),
    ("mi355_scan2_dev", _int, [_vp, _vp, C.c_uint, _int, C.c_int64, C.c_int64, _vp, C.c_uint, _int, C.c_int64, C.c_int64, _u64, _int,
                         _vp, _vp]),
    ("mi355_scan_select_dev", _int, [_vp, _vp, _u64, C.c_uint, _int, C.c_int64, C.c_int64, _int, _vp, _u64, _vp, _u64, _vp]),
    ("mi355_scan_in_dev", _int, [_vp, _vp, _u64, C.c_uint, _vp, C.c_uint, _int, _vp, _vp, _vp]),
    ("mi355_bitmap_combine_dev", _int, [_vp, _int, _vp, _vp, _vp, _u64, _vp]),
    ("mi355_bitmap_count_dev", _int, [_vp, _vp, _u64, _vp]),
    ("mi355_bitmap_to_rowids_dev", _int, [_vp, _vp, _u64, _u64, _vp, _u64, _vp]),
    ("mi355_gather_dev", _int, [_vp, _vp, _u64, C.c_uint, _u64, _vp, _vp, _u64, _vp]),
    ("mi355_aggregate_dev", _int, [_vp, _vp, _u64, C.c_uint, _vp, _vp]),
    ("mi355_histogram_dev", _int, [_vp, _vp, _u64, C.c_uint, _vp, _vp]),
    ("mi355_comm_get_unique_id", _int, [_vp]),
    ("mi355_comm_create", _int, [_vp, _int, _int, _vp, C.POINTER(_vp)]),
    ("mi355_comm_destroy", _int, [_vp]),
    ("mi355_comm_info", _int, [_vp, C.POINTER(_int), C.POINTER(_int)]),
    ("mi355_gather_bitmaps_dev", _int, [_vp, _vp, _vp, _vp, _int, _vp]),
    ("mi355_gather_bitmaps_at_dev", _int, [_vp, _vp, _vp, _vp, _vp, _int, _vp]),
    ("mi355_allreduce_hits_dev", _int, [_vp, _vp, _vp, C.c_uint]),
    ("mi355_sharded_scan_eq_dev", _int, [_vp, _vp, _vp, C.c_uint, _i32, _vp, _vp, _int, _vp, _vp]),
    ("mi355_sharded_scan_range_dev", _int, [_vp, _vp, _vp, C.c_uint, _u32, _u32, _vp, _vp, _int, _vp, _vp]),
    ("mi355_kernel_name", C.c_char_p, [C.c_char_p, C.c_uint]),
    ("mi355_shared_scan_kernel", C.c_char_p, [_vp, C.c_uint, C.c_uint, _int, _int]),
    ("mi355_tile_values", _u64, [C.c_uint]),
]


class Mi355Error(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load shared_simd_scan_amd/libmi355scan.so (built by build.py / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mi355Error(
                f"{LIB_PATH} not found: the HIP extension is not built (run `python -m shared_simd_scan_amd.build`). "
                "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        msg = lib().mi355_last_error()
        raise Mi355Error(f"mi355 error {rc}: {msg.decode() if msg else '?'}")

"""ScanEngine -- Python face of the C ABI for device-resident columns.

torch is plumbing here: it owns device memory (uint8 / int32 tensors), the HIP stream the kernels are
enqueued on, and (in sharded.py) the RCCL process group.  Every operation is one call into
libmi355scan.so; nothing is computed in Python and nothing falls back to the CPU.

Names follow the reference's surface (src/simd_scan.hpp): compress / decompress / scan / shared_scan.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi
from ._capi import check, lib


def compressed_buffer_size(c: int, n: int) -> int:
    """src/simd_scan.hpp:20-26"""
    return lib().mi355_compressed_buffer_size(c, n)


def decompression_output_buffer_size(n: int) -> int:
    """src/simd_scan.hpp:28-33"""
    return lib().mi355_decompression_output_buffer_size(n)


def scan_output_buffer_size(n: int) -> int:
    """src/simd_scan.hpp:35-40"""
    return lib().mi355_scan_output_buffer_size(n)


def tile_values(c: int) -> int:
    return int(lib().mi355_tile_values(c))


def kernel_name(op: str, c: int) -> str:
    s = lib().mi355_kernel_name(op.encode(), c)
    if s is None:
        raise ValueError(op)
    return s.decode()


class PackedColumn:
    """A bit-packed column resident in HBM: `n` values of `c` bits, reference stream format."""

    def __init__(self, data: torch.Tensor, n: int, c: int):
        assert data.dtype == torch.uint8 and data.is_cuda
        self.data, self.n, self.c = data, int(n), int(c)

    @property
    def payload_bytes(self) -> int:
        return (self.n * self.c + 7) // 8


class ScanEngine:
    def __init__(self, device: Optional[int] = None, stream: Optional[torch.cuda.Stream] = None):
        if not torch.cuda.is_available():
            raise _capi.Mi355Error("no GPU visible: shared_simd_scan_amd has no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        with torch.cuda.device(self.device):
            self.stream = stream if stream is not None else torch.cuda.current_stream()
        self._ctx = C.c_void_p()
        check(lib().mi355_ctx_create(self.device, C.c_void_p(self.stream.cuda_stream), C.byref(self._ctx)))
        self._dev = torch.device("cuda", self.device)

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx:
            lib().mi355_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int) -> None:
        check(lib().mi355_ctx_set_option(self._ctx, name.encode(), int(value)))

    TUNE_KINDS = {"scan": 1, "count": 2, "mask": 4, "decompress": 8, "all": 15}

    def tune(self, col: "PackedColumn", what: str = "all") -> dict:
        """mi355_tune_dev: measure 1 / 2 / 4 resident blocks per CU on this column (>= 5e7 rows; blocks) and keep the
        fastest for later launches of that kind and width.  Returns what was kept ({} for a small column)."""
        mask = 0
        for kind in what.split("+"):
            mask |= self.TUNE_KINDS[kind]
        check(lib().mi355_tune_dev(self._ctx, col.data.data_ptr(), col.n, col.c, mask))
        return self.tuned(col.c)

    def tuned(self, c: int) -> dict:
        got = {"scan_eq": lib().mi355_tuned_blocks_per_cu(self._ctx, c, 1, 0), "scan_range": lib().mi355_tuned_blocks_per_cu(self._ctx, c, 1, 1),
               "count": lib().mi355_tuned_blocks_per_cu(self._ctx, c, 2, 0), "mask": lib().mi355_tuned_blocks_per_cu(self._ctx, c, 4, 0),
               "decompress": lib().mi355_tuned_blocks_per_cu(self._ctx, c, 8, 0)}
        return {k: v for k, v in got.items() if v}

    def use_stream(self, stream: torch.cuda.Stream) -> None:
        """enqueue subsequent work on `stream` (e.g. the capture stream inside torch.cuda.graph)"""
        self.stream = stream
        check(lib().mi355_ctx_set_stream(self._ctx, C.c_void_p(stream.cuda_stream)))

    def synchronize(self) -> None:
        check(lib().mi355_ctx_synchronize(self._ctx))

    # ---- allocation -----------------------------------------------------------------------
    def _empty(self, nbytes: int, dtype=torch.uint8) -> torch.Tensor:
        return torch.empty(nbytes, dtype=dtype, device=self._dev)

    def alloc_packed(self, n: int, c: int) -> PackedColumn:
        return PackedColumn(self._empty(compressed_buffer_size(c, n)), n, c)

    # ---- compression (src/simd_scan_compression.cpp:53-104) ----------------------------------
    def compress(self, values: torch.Tensor, c: int) -> PackedColumn:
        """values: uint16 / int32 / uint32-as-int32 tensor on this device -> packed column."""
        values = values.to(self._dev).contiguous()
        n = values.numel()
        col = self.alloc_packed(n, c)
        if values.dtype in (torch.uint16, torch.int16):
            check(lib().mi355_pack_u16_dev(self._ctx, values.data_ptr(), n, c, col.data.data_ptr()))
        elif values.dtype in (torch.int32, torch.uint32):
            check(lib().mi355_pack_u32_dev(self._ctx, values.data_ptr(), n, c, col.data.data_ptr()))
        else:
            raise TypeError(f"values dtype {values.dtype}: need a 16- or 32-bit integer tensor")
        return col

    def slice_rows(self, col: PackedColumn, first: int, last: int) -> PackedColumn:
        """View of rows [first, last) of a resident column (no copy).  `first` must be a multiple of 128 rows so the
        slice starts on a whole value, 16-byte aligned; a slice that ends inside the column is followed by the next
        rows instead of the zero pad, which only feeds result bits >= n (masked by every kernel)."""
        if first % 128 or not (0 <= first <= last <= col.n):
            raise ValueError("slice_rows: first must be a multiple of 128 and 0 <= first <= last <= n")
        return PackedColumn(col.data[first * col.c // 8:], last - first, col.c)

    def generate(self, kind: str, n: int, c: int, param: int = 0, first_row: int = 0) -> PackedColumn:
        """Synthesise a packed column on the device: 'mod' | 'splitmix' | 'index' (SURVEY 8d)."""
        code = {"mod": _capi.GEN_MOD, "splitmix": _capi.GEN_SPLITMIX, "index": _capi.GEN_INDEX}[kind]
        col = self.alloc_packed(n, c)
        check(lib().mi355_generate_dev(self._ctx, code, first_row, n, c, param, col.data.data_ptr()))
        return col

    # ---- decompression (src/simd_scan_decompression.cpp) --------------------------------------
    def decompress(self, col: PackedColumn, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = self._empty(col.n, torch.int32)
        assert out.dtype == torch.int32 and out.numel() >= col.n
        check(lib().mi355_decompress_dev(self._ctx, col.data.data_ptr(), col.n, col.c, out.data_ptr()))
        return out

    # ---- scans (src/simd_scan.cpp) -------------------------------------------------------------
    def alloc_bitmap(self, n: int) -> torch.Tensor:
        return self._empty((n + 7) // 8)

    def scan(self, key: int, col: PackedColumn, bitmap: Optional[torch.Tensor] = None,
             hits: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """value == key -> (bitmap uint8[ceil(n/8)], hits int64[1]); asynchronous on the stream."""
        if bitmap is None:
            bitmap = self.alloc_bitmap(col.n)
        if hits is None:
            hits = torch.empty(1, dtype=torch.int64, device=self._dev)
        key32 = int(np.int32(np.uint32(int(key) & 0xFFFFFFFF)))
        check(lib().mi355_scan_eq_dev(self._ctx, col.data.data_ptr(), col.n, col.c, key32, bitmap.data_ptr(),
                                      hits.data_ptr()))
        return bitmap, hits

    def scan_range(self, lo: int, hi: int, col: PackedColumn, bitmap: Optional[torch.Tensor] = None,
                   hits: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """lo <= value <= hi (src/simd_scan.hpp:76-84)."""
        if bitmap is None:
            bitmap = self.alloc_bitmap(col.n)
        if hits is None:
            hits = torch.empty(1, dtype=torch.int64, device=self._dev)
        check(lib().mi355_scan_range_dev(self._ctx, col.data.data_ptr(), col.n, col.c, lo, hi, bitmap.data_ptr(),
                                         hits.data_ptr()))
        return bitmap, hits

    # ---- beyond the reference: general comparisons, conjunctions, bitmap consumers -----
    _CMP = {"==": 0, "!=": 1, "<": 2, "<=": 3, ">": 4, ">=": 5, "between": 6, "not_between": 7}
    _BOP = {"and": 0, "or": 1, "xor": 2, "andnot": 3}

    def scan_where(self, op: str, a: int, col: PackedColumn, b: int = 0, and_mask: Optional[torch.Tensor] = None,
                   bitmap: Optional[torch.Tensor] = None, hits: Optional[torch.Tensor] = None):
        """bitmap[i] = (value_i OP a [, b]) [& and_mask[i]]; op in == != < <= > >= between not_between."""
        if bitmap is None:
            bitmap = self.alloc_bitmap(col.n)
        if hits is None:
            hits = torch.empty(1, dtype=torch.int64, device=self._dev)
        check(lib().mi355_scan_where_dev(self._ctx, col.data.data_ptr(), col.n, col.c, self._CMP[op], int(a), int(b),
                                         and_mask.data_ptr() if and_mask is not None else None, bitmap.data_ptr(),
                                         hits.data_ptr()))
        return bitmap, hits

    def scan_combine(self, op: str, a: int, col: PackedColumn, b: int = 0, mask: Optional[torch.Tensor] = None,
                     mask_op: str = "and", bitmap: Optional[torch.Tensor] = None, hits: Optional[torch.Tensor] = None,
                     count_only: bool = False):
        """scan_where with the earlier bitmap combined by AND / OR / XOR / ANDNOT (mask & ~p) inside the scan;
        count_only=True stores no bitmap at all (returns (None, hits))."""
        if bitmap is None and not count_only:
            bitmap = self.alloc_bitmap(col.n)
        if hits is None:
            hits = torch.empty(1, dtype=torch.int64, device=self._dev)
        check(lib().mi355_scan_combine_dev(self._ctx, col.data.data_ptr(), col.n, col.c, self._CMP[op], int(a), int(b),
                                           self._BOP[mask_op], mask.data_ptr() if mask is not None else None,
                                           None if count_only else bitmap.data_ptr(), hits.data_ptr()))
        return (None if count_only else bitmap), hits

    def scan2(self, col1: PackedColumn, op1: str, a1: int, col2: PackedColumn, op2: str, a2: int, b1: int = 0, b2: int = 0,
              combine: str = "and", bitmap: Optional[torch.Tensor] = None, hits: Optional[torch.Tensor] = None,
              count_only: bool = False):
        """predicates over two columns of the same row count, combined (and / or / xor / andnot = p1 & ~p2) in one call;
        same-width columns run as a single launch"""
        assert col1.n == col2.n
        if bitmap is None and not count_only:
            bitmap = self.alloc_bitmap(col1.n)
        if hits is None:
            hits = torch.empty(1, dtype=torch.int64, device=self._dev)
        check(lib().mi355_scan2_dev(self._ctx, col1.data.data_ptr(), col1.c, self._CMP[op1], int(a1), int(b1), col2.data.data_ptr(),
                                    col2.c, self._CMP[op2], int(a2), int(b2), col1.n, self._BOP[combine],
                                    None if count_only else bitmap.data_ptr(), hits.data_ptr()))
        return (None if count_only else bitmap), hits

    def scan_select(self, op: str, a: int, col: PackedColumn, capacity: int, b: int = 0, mask: Optional[torch.Tensor] = None,
                    mask_op: str = "and", first_row: int = 0):
        """predicate (optionally combined with an earlier bitmap) -> (int64 ascending row ids [capacity], count int64[1])
        in one launch, no bitmap in HBM; ids beyond `capacity` are dropped, count is the total."""
        rowids = torch.empty(max(capacity, 1), dtype=torch.int64, device=self._dev)
        count = torch.empty(1, dtype=torch.int64, device=self._dev)
        check(lib().mi355_scan_select_dev(self._ctx, col.data.data_ptr(), col.n, col.c, self._CMP[op], int(a), int(b),
                                          self._BOP[mask_op], mask.data_ptr() if mask is not None else None, first_row,
                                          rowids.data_ptr(), capacity, count.data_ptr()))
        return rowids, count

    def scan_in(self, keys: Sequence[int], col: PackedColumn, negate: bool = False,
                and_mask: Optional[torch.Tensor] = None, bitmap: Optional[torch.Tensor] = None,
                hits: Optional[torch.Tensor] = None):
        """bitmap[i] = value_i in keys (NOT IN with negate=True) [& and_mask[i]]."""
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.int64).astype(np.int32))
        if bitmap is None:
            bitmap = self.alloc_bitmap(col.n)
        if hits is None:
            hits = torch.empty(1, dtype=torch.int64, device=self._dev)
        check(lib().mi355_scan_in_dev(self._ctx, col.data.data_ptr(), col.n, col.c, k.ctypes.data, int(k.shape[0]),
                                      1 if negate else 0, and_mask.data_ptr() if and_mask is not None else None,
                                      bitmap.data_ptr(), hits.data_ptr()))
        return bitmap, hits

    def bitmap_combine(self, op: str, a: torch.Tensor, b: torch.Tensor, n: int, out: Optional[torch.Tensor] = None):
        if out is None:
            out = self.alloc_bitmap(n)
        count = torch.empty(1, dtype=torch.int64, device=self._dev)
        check(lib().mi355_bitmap_combine_dev(self._ctx, self._BOP[op], a.data_ptr(), b.data_ptr(), out.data_ptr(), n,
                                             count.data_ptr()))
        return out, count

    def bitmap_count(self, bitmap: torch.Tensor, n: int) -> torch.Tensor:
        count = torch.empty(1, dtype=torch.int64, device=self._dev)
        check(lib().mi355_bitmap_count_dev(self._ctx, bitmap.data_ptr(), n, count.data_ptr()))
        return count

    def bitmap_to_rowids(self, bitmap: torch.Tensor, n: int, capacity: int, first_row: int = 0):
        """-> (int64 row ids [capacity], count int64[1]); ids beyond `capacity` are dropped, count is the total."""
        rowids = torch.empty(max(capacity, 1), dtype=torch.int64, device=self._dev)
        count = torch.empty(1, dtype=torch.int64, device=self._dev)
        check(lib().mi355_bitmap_to_rowids_dev(self._ctx, bitmap.data_ptr(), n, first_row, rowids.data_ptr(), capacity,
                                               count.data_ptr()))
        return rowids, count

    def gather(self, col: PackedColumn, rowids: torch.Tensor, count, first_row: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """values of `col` at the row ids a selection produced ("take"): out[i] = value of row rowids[i] for
        i < min(count, len(rowids)); `count` is the int64[1] device tensor the selection left behind (no host round trip) or
        an int.  Ids outside the column give -1."""
        cap = int(rowids.numel())
        if not torch.is_tensor(count):
            count = torch.tensor([int(count)], dtype=torch.int64, device=self._dev)
        if out is None:
            out = torch.empty(max(cap, 1), dtype=torch.int32, device=self._dev)
        assert rowids.dtype == torch.int64 and out.dtype == torch.int32 and out.numel() >= cap
        check(lib().mi355_gather_dev(self._ctx, col.data.data_ptr(), col.n, col.c, first_row, rowids.data_ptr(), count.data_ptr(), cap,
                                     out.data_ptr()))
        return out

    def aggregate(self, col: PackedColumn, mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> int64[4] device tensor (sum, count, min, max) of the column's values over the rows of `mask` (a result bitmap;
        None = every row): one pass over the column, nothing decoded to memory.  count = 0: min is -1 (UINT64_MAX)."""
        if out is None:
            out = torch.empty(4, dtype=torch.int64, device=self._dev)
        check(lib().mi355_aggregate_dev(self._ctx, col.data.data_ptr(), col.n, col.c, mask.data_ptr() if mask is not None else None,
                                        out.data_ptr()))
        return out

    def histogram(self, col: PackedColumn, mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> int64[2^c] device tensor: how often each value occurs among the rows of `mask` (None = all rows); c <= 14"""
        if out is None:
            out = torch.empty(1 << col.c, dtype=torch.int64, device=self._dev)
        assert out.numel() >= (1 << col.c) and out.dtype == torch.int64
        check(lib().mi355_histogram_dev(self._ctx, col.data.data_ptr(), col.n, col.c, mask.data_ptr() if mask is not None else None,
                                        out.data_ptr()))
        return out

    # ---- shared scans (src/simd_scan_shared.cpp, src/simd_scan_shared_linear.cpp) -----------------
    def shared_scan(self, keys: Sequence[int], col: PackedColumn, layout: str = "per_predicate",
                    out: Optional[torch.Tensor] = None, hits: Optional[torch.Tensor] = None):
        """per_predicate -> uint8[P, stride] (row k = bitmap of keys[k], stride = mi355_bitmap_stride(n): ceil(n/8) rounded up to 256);
        linear -> uint8[ceil(n/8) * P] with the byte of 8-value group g and key k at g*P + k.
        hits: int64[P] device tensor to fill (allocated when None); False skips the hit counts."""
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.int64).astype(np.int32))
        P = int(k.shape[0])
        nb = (col.n + 7) // 8
        if hits is None:
            hits = torch.empty(P, dtype=torch.int64, device=self._dev)
        hits_ptr = 0 if hits is False else hits.data_ptr()  # hits=False: bitmaps only, no counting
        if layout == "per_predicate":
            stride = int(lib().mi355_bitmap_stride(col.n))  # whole 128-byte lines per bitmap: up to 2x faster than a 16-byte-multiple stride
            if out is None:
                out = torch.empty((P, stride), dtype=torch.uint8, device=self._dev)
            code = _capi.LAYOUT_PER_PREDICATE
        elif layout == "linear":
            stride = 0
            if out is None:
                out = self._empty(nb * P)
            code = _capi.LAYOUT_LINEAR
        else:
            raise ValueError(layout)
        check(lib().mi355_shared_scan_eq_dev(self._ctx, col.data.data_ptr(), col.n, col.c, k.ctypes.data, P, code,
                                             out.data_ptr(), stride, hits_ptr))
        return out, (None if hits is False else hits)

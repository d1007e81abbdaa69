"""oracle -- TEST INFRASTRUCTURE (never imported by shared_simd_scan_amd).

ctypes doors onto
  * ``liboracle.so``   : the plain-C restatement of the reference's path (oracle.c), and
  * ``_ref/libref_w<W>.so`` : the reference itself, compiled from its own sources by
    ``oracle/Makefile`` in the build container (absent files simply make ``RefLib`` raise).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the checker.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_u8p = C.POINTER(C.c_uint8)
_sz = C.c_size_t


def build(ref: bool = True) -> None:
    """Compile liboracle.so (and, when /root/reference is present, oracle/_ref/*.so)."""
    target = "all" if ref else "oracle"
    subprocess.run(["make", "-C", _HERE, "-j8", target], check=True, stdout=subprocess.DEVNULL)


def _ptr(a: np.ndarray, ty=C.c_void_p):
    return a.ctypes.data_as(ty)


class _Oracle:
    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.oracle_compressed_buffer_size.restype = _sz
        L.oracle_compressed_buffer_size.argtypes = [C.c_uint, _sz]
        L.oracle_decompression_output_buffer_size.restype = _sz
        L.oracle_decompression_output_buffer_size.argtypes = [_sz]
        L.oracle_scan_output_buffer_size.restype = _sz
        L.oracle_scan_output_buffer_size.argtypes = [_sz]
        L.oracle_get_bit.restype = C.c_int
        L.oracle_get_bit.argtypes = [C.c_void_p, _sz]
        L.oracle_pack_u16.argtypes = [C.c_void_p, _sz, C.c_uint, C.c_void_p]
        L.oracle_pack_u32.argtypes = [C.c_void_p, _sz, C.c_uint, C.c_void_p]
        L.oracle_decompress.argtypes = [C.c_void_p, _sz, C.c_uint, C.c_void_p]
        L.oracle_scan_eq.restype = C.c_uint64
        L.oracle_scan_eq.argtypes = [C.c_void_p, _sz, C.c_uint, C.c_int32, C.c_void_p]
        L.oracle_scan_range.restype = C.c_uint64
        L.oracle_scan_range.argtypes = [C.c_void_p, _sz, C.c_uint, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_shared_scan_eq.argtypes = [C.c_void_p, _sz, C.c_uint, C.c_void_p, _sz, C.c_int, C.c_void_p, _sz,
                                            C.c_void_p]
        L.oracle_gen_values.argtypes = [C.c_int, C.c_uint64, _sz, C.c_uint, C.c_uint64, C.c_void_p]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_avx2_available.restype = C.c_int
        L.oracle_avx2_scan_eq.restype = C.c_uint64
        L.oracle_avx2_scan_eq.argtypes = [C.c_void_p, _sz, C.c_uint, C.c_int32, C.c_void_p, C.c_int]
        L.oracle_set_num_threads.argtypes = [C.c_int]
        self.L = L

    # -- sizing -----------------------------------------------------------------------
    def compressed_buffer_size(self, c: int, n: int) -> int:
        return self.L.oracle_compressed_buffer_size(c, n)

    def decompression_output_buffer_size(self, n: int) -> int:
        return self.L.oracle_decompression_output_buffer_size(n)

    def scan_output_buffer_size(self, n: int) -> int:
        return self.L.oracle_scan_output_buffer_size(n)

    def get_bit(self, bitmap: np.ndarray, i: int) -> bool:
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
        return bool(self.L.oracle_get_bit(_ptr(bitmap), i))

    # -- path -------------------------------------------------------------------------
    def pack(self, values: np.ndarray, c: int) -> np.ndarray:
        """-> uint8[compressed_buffer_size(c, n)] (zero padded), reference format."""
        n = int(values.shape[0])
        out = np.zeros(self.compressed_buffer_size(c, n) + 8, dtype=np.uint8)
        if values.dtype == np.uint16:
            v = np.ascontiguousarray(values)
            self.L.oracle_pack_u16(_ptr(v), n, c, _ptr(out))
        else:
            v = np.ascontiguousarray(values, dtype=np.uint32)
            self.L.oracle_pack_u32(_ptr(v), n, c, _ptr(out))
        return out[: self.compressed_buffer_size(c, n)]

    def decompress(self, packed: np.ndarray, n: int, c: int) -> np.ndarray:
        packed = self._padded(packed, n, c)
        out = np.empty(n, dtype=np.int32)
        self.L.oracle_decompress(_ptr(packed), n, c, _ptr(out))
        return out

    def scan_eq(self, packed: np.ndarray, n: int, c: int, key: int):
        packed = self._padded(packed, n, c)
        out = np.zeros((n + 7) // 8, dtype=np.uint8)
        hits = self.L.oracle_scan_eq(_ptr(packed), n, c, int(np.int32(np.uint32(key & 0xFFFFFFFF))), _ptr(out))
        return out, int(hits)

    def avx2_available(self) -> bool:
        return bool(self.L.oracle_avx2_available())

    def scan_eq_avx2(self, packed: np.ndarray, n: int, c: int, key: int, threads: int = 1, reps: int = 1):
        """oracle_avx2.c: the AVX2 (+ OpenMP when threads != 1; 0 = all cores) restatement of the reference's
        scan_256_unrolled.  Needs 16 readable bytes past the payload (the reference's 256-byte pad).
        -> (bitmap uint8[ceil(n/8)], hits, seconds per rep as a list)"""
        import time

        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        need = (n * c + 7) // 8 + 32
        if packed.shape[0] < need:
            p = np.zeros(need, dtype=np.uint8)
            p[: packed.shape[0]] = packed
            packed = p
        out = np.zeros((n + 7) // 8, dtype=np.uint8)
        key32 = int(np.int32(np.uint32(key & 0xFFFFFFFF)))
        secs, hits = [], 0
        for _ in range(max(1, reps)):
            t0 = time.perf_counter()
            hits = self.L.oracle_avx2_scan_eq(_ptr(packed), n, c, key32, _ptr(out), threads)
            secs.append(time.perf_counter() - t0)
        return out, int(hits), secs

    def scan_range(self, packed: np.ndarray, n: int, c: int, lo: int, hi: int):
        packed = self._padded(packed, n, c)
        out = np.zeros((n + 7) // 8, dtype=np.uint8)
        hits = self.L.oracle_scan_range(_ptr(packed), n, c, lo, hi, _ptr(out))
        return out, int(hits)

    def shared_scan_eq(self, packed: np.ndarray, n: int, c: int, keys, layout: str = "per_predicate"):
        """per_predicate -> uint8[P, ceil(n/8)];  linear -> uint8[ceil(n/8) * P] (byte g*P+k)."""
        packed = self._padded(packed, n, c)
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.int64).astype(np.int32))
        P = int(k.shape[0])
        nb = (n + 7) // 8
        hits = np.zeros(P, dtype=np.uint64)
        if layout == "per_predicate":
            out = np.zeros((P, nb), dtype=np.uint8)
            self.L.oracle_shared_scan_eq(_ptr(packed), n, c, _ptr(k), P, 0, _ptr(out), nb, _ptr(hits))
        else:
            out = np.zeros(nb * P, dtype=np.uint8)
            self.L.oracle_shared_scan_eq(_ptr(packed), n, c, _ptr(k), P, 1, _ptr(out), 0, _ptr(hits))
        return out, hits

    def gen_values(self, kind: str, n: int, c: int, param: int = 0, first: int = 0) -> np.ndarray:
        """kind: 'mod' (v=(first+i)%param), 'splitmix' (seed=param), 'index' (v=(first+i)&mask)."""
        code = {"mod": 0, "splitmix": 1, "index": 2}[kind]
        out = np.empty(n, dtype=np.uint32)
        self.L.oracle_gen_values(code, first, n, c, param, _ptr(out))
        return out

    def num_threads(self) -> int:
        return self.L.oracle_num_threads()

    def set_num_threads(self, t: int) -> None:
        self.L.oracle_set_num_threads(t)

    # the C code reads up to 5 bytes from floor(c*i/8): make sure the tail is there
    def _padded(self, packed: np.ndarray, n: int, c: int) -> np.ndarray:
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        need = (n * c + 7) // 8 + 8
        if packed.shape[0] < need:
            p = np.zeros(need, dtype=np.uint8)
            p[: packed.shape[0]] = packed
            return p
        return packed


_oracle = None


def oracle() -> _Oracle:
    global _oracle
    if _oracle is None:
        _oracle = _Oracle()
    return _oracle


DECOMP_VARIANTS = {
    "decompress_unvectorized": 0, "decompress_128_sweep": 1, "decompress_128_nosweep": 2, "decompress_128_9bit": 3,
    "decompress_128": 4, "decompress_128_unrolled": 5, "decompress_128_aligned": 6, "decompress_256": 7,
    "decompress_256_avx2": 8,
}
SCAN_VARIANTS = {"scan_unvectorized": 0, "scan_128": 1, "scan_128_unrolled": 2, "scan_256": 3, "scan_256_unrolled": 4}
SHARED_VARIANTS = {
    "shared_scan_128_sequential": 0, "shared_scan_128_sequential_unrolled": 1, "shared_scan_128_threaded": 2,
    "shared_scan_128_standard": 3, "shared_scan_128_standard_unrolled": 4, "shared_scan_128_parallel": 5,
    "shared_scan_256_sequential": 6, "shared_scan_256_standard": 7, "shared_scan_256_parallel": 8,
}
LINEAR_VARIANTS = {"shared_scan_128_linear_standard": 0, "shared_scan_128_linear_simple": 1}


def ref_available(width: int = 9) -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", f"libref_w{width}.so"))


class RefLib:
    """The reference itself at one compile-time width (oracle/_ref/libref_w<W>.so)."""

    def __init__(self, width: int = 9):
        path = os.path.join(_HERE, "_ref", f"libref_w{width}.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: build with `make -C oracle ref` where /root/reference exists")
        L = C.CDLL(path)
        L.ref_width.restype = C.c_int
        for f in ("ref_compressed_buffer_size", "ref_decompression_output_buffer_size", "ref_scan_output_buffer_size"):
            getattr(L, f).restype = _sz
            getattr(L, f).argtypes = [_sz]
        L.ref_next_multiple.restype = C.c_int
        L.ref_next_multiple.argtypes = [C.c_int, C.c_int]
        L.ref_get_bit.restype = C.c_int
        L.ref_get_bit.argtypes = [C.c_void_p, _sz, _sz]
        L.ref_compress.argtypes = [C.c_void_p, _sz, C.c_void_p]
        L.ref_decompress.restype = C.c_int
        L.ref_decompress.argtypes = [C.c_int, C.c_void_p, _sz, C.c_void_p]
        L.ref_scan.restype = C.c_int
        L.ref_scan.argtypes = [C.c_int, C.c_int, C.c_void_p, _sz, C.c_void_p]
        L.ref_scan_timed.restype = C.c_int
        L.ref_scan_timed.argtypes = [C.c_int, C.c_int, C.c_void_p, _sz, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_shared_scan.restype = C.c_int
        L.ref_shared_scan.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, _sz, C.c_void_p]
        L.ref_shared_scan_timed.restype = C.c_int
        L.ref_shared_scan_timed.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, _sz, C.c_int, C.c_void_p]
        L.ref_shared_scan_linear.restype = C.c_int
        L.ref_shared_scan_linear.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, _sz, C.c_void_p]
        L.ref_decompress_timed.restype = C.c_int
        L.ref_decompress_timed.argtypes = [C.c_int, C.c_void_p, _sz, C.c_int, C.c_void_p, C.c_void_p]
        self.L = L
        self.width = width
        assert L.ref_width() == width

    def compressed_buffer_size(self, n):
        return self.L.ref_compressed_buffer_size(n)

    def scan_output_buffer_size(self, n):
        return self.L.ref_scan_output_buffer_size(n)

    def decompression_output_buffer_size(self, n):
        return self.L.ref_decompression_output_buffer_size(n)

    def next_multiple(self, a, b):
        return self.L.ref_next_multiple(a, b)

    def get_bit(self, bitmap: np.ndarray, i: int) -> bool:
        b = np.ascontiguousarray(bitmap, dtype=np.uint8)
        return bool(self.L.ref_get_bit(_ptr(b), b.shape[0], i))

    def compress(self, values_u16: np.ndarray) -> np.ndarray:
        v = np.ascontiguousarray(values_u16, dtype=np.uint16)
        n = v.shape[0]
        out = np.zeros(self.compressed_buffer_size(n), dtype=np.uint8)
        self.L.ref_compress(_ptr(v), n, _ptr(out))
        return out

    def _packed(self, packed):
        # every SIMD variant may read 16 B past the last packed byte: keep the 256 B pad
        return np.ascontiguousarray(packed, dtype=np.uint8)

    def decompress(self, variant: str, packed: np.ndarray, n: int) -> np.ndarray:
        packed = self._packed(packed)
        out = np.zeros(self.decompression_output_buffer_size(n) // 4 + 64, dtype=np.int32)
        rc = self.L.ref_decompress(DECOMP_VARIANTS[variant], _ptr(packed), n, _ptr(out))
        assert rc == 0
        return out

    def scan(self, variant: str, key: int, packed: np.ndarray, n: int):
        """-> (whole padded output buffer, the reference's int hits)"""
        packed = self._packed(packed)
        out = np.zeros(self.scan_output_buffer_size(n), dtype=np.uint8)
        hits = self.L.ref_scan(SCAN_VARIANTS[variant], int(key), _ptr(packed), n, _ptr(out))
        return out, hits

    def scan_timed(self, variant: str, key: int, packed: np.ndarray, n: int, reps: int):
        packed = self._packed(packed)
        secs = np.zeros(reps, dtype=np.float64)
        out = np.zeros(self.scan_output_buffer_size(n), dtype=np.uint8)
        hits = self.L.ref_scan_timed(SCAN_VARIANTS[variant], int(key), _ptr(packed), n, reps, _ptr(secs), _ptr(out))
        return secs, out, hits

    def shared_scan(self, variant: str, keys, packed: np.ndarray, n: int) -> np.ndarray:
        packed = self._packed(packed)
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.int32))
        P = k.shape[0]
        out = np.zeros((P, self.scan_output_buffer_size(n)), dtype=np.uint8)
        rc = self.L.ref_shared_scan(SHARED_VARIANTS[variant], _ptr(k), P, _ptr(packed), n, _ptr(out))
        assert rc == 0
        return out

    def shared_scan_timed(self, variant: str, keys, packed: np.ndarray, n: int, reps: int):
        packed = self._packed(packed)
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.int32))
        secs = np.zeros(reps, dtype=np.float64)
        rc = self.L.ref_shared_scan_timed(SHARED_VARIANTS[variant], _ptr(k), k.shape[0], _ptr(packed), n, reps,
                                          _ptr(secs))
        assert rc == 0
        return secs

    def shared_scan_linear(self, variant: str, keys, packed: np.ndarray, n: int) -> np.ndarray:
        packed = self._packed(packed)
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.int32))
        P = k.shape[0]
        out = np.zeros(P * self.scan_output_buffer_size(n), dtype=np.uint8)
        rc = self.L.ref_shared_scan_linear(LINEAR_VARIANTS[variant], _ptr(k), P, _ptr(packed), n, _ptr(out))
        if rc != 0:
            raise NotImplementedError(variant)
        return out

    def decompress_timed(self, variant: str, packed: np.ndarray, n: int, reps: int):
        packed = self._packed(packed)
        out = np.zeros(self.decompression_output_buffer_size(n) // 4 + 64, dtype=np.int32)
        secs = np.zeros(reps, dtype=np.float64)
        rc = self.L.ref_decompress_timed(DECOMP_VARIANTS[variant], _ptr(packed), n, reps, _ptr(secs), _ptr(out))
        assert rc == 0
        return secs, out

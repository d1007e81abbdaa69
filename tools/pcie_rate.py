#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer (drop-in) entry point mi355_scan_eq: packed column in host memory in,
bitmap in host memory out.  Never the bench `value`; quoted in DESIGN.md section 5."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shared_simd_scan_amd import lib  # noqa: E402

L = lib()
n, c = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000, 9
nbytes = L.mi355_compressed_buffer_size(c, n)
packed = np.zeros(nbytes, dtype=np.uint8)
packed[: nbytes - 256] = np.random.default_rng(0).integers(0, 256, size=nbytes - 256, dtype=np.uint8)
out = np.zeros(L.mi355_scan_output_buffer_size(n), dtype=np.uint8)
hits = C.c_uint64()
for i in range(3):
    t0 = time.perf_counter()
    rc = L.mi355_scan_eq(None, packed.ctypes.data, n, c, 3, out.ctypes.data, C.byref(hits))
    dt = time.perf_counter() - t0
    assert rc == 0
    print(f"host-pointer scan_eq n={n}: {dt * 1e3:.1f} ms  -> {n / dt:.3e} values/s ({(nbytes + n / 8) / dt / 1e9:.1f} GB/s over PCIe, pageable host memory), hits={hits.value}")

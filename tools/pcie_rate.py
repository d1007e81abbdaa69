#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer (drop-in) entry point mi355_scan_eq: packed column in host memory in,
bitmap in host memory out.  Never the bench `value`; quoted in DESIGN.md section 5.

    python tools/pcie_rate.py [rows ...]     (default: 1e4 1e6 1e7 1e8 1e9)
Per size: the first call (device buffers of the context's pool are allocated, the runtime pins the host pages) and the
best / median of the following calls."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shared_simd_scan_amd import lib  # noqa: E402

L = lib()
c = 9
sizes = [int(float(a)) for a in sys.argv[1:]] or [10_000, 1_000_000, 10_000_000, 100_000_000, 1_000_000_000]
for n in sizes:
    nbytes = L.mi355_compressed_buffer_size(c, n)
    packed = np.zeros(nbytes, dtype=np.uint8)
    packed[: nbytes - 256] = np.random.default_rng(0).integers(0, 256, size=nbytes - 256, dtype=np.uint8)
    out = np.zeros(L.mi355_scan_output_buffer_size(n), dtype=np.uint8)
    hits = C.c_uint64()
    ts = []
    for i in range(3 if n >= 100_000_000 else 30):
        t0 = time.perf_counter()
        rc = L.mi355_scan_eq(None, packed.ctypes.data, n, c, 3, out.ctypes.data, C.byref(hits))
        ts.append(time.perf_counter() - t0)
        assert rc == 0, L.mi355_last_error()
    rest = sorted(ts[1:])
    best, med = rest[0], rest[len(rest) // 2]
    print(f"host-pointer scan_eq n={n:>10}: first {ts[0] * 1e3:9.3f} ms, then best {best * 1e3:9.3f} / median {med * 1e3:9.3f} ms"
          f" -> {n / best:.3e} values/s ({(nbytes + n / 8) / best / 1e9:.1f} GB/s over PCIe, pageable host memory), hits={hits.value}",
          flush=True)

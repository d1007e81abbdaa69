#!/usr/bin/env python3
"""IN-list scan with and without a fused AND-mask, 1e9 x 9 bit, launches back to back (the mask travels by LDS-DMA with the tile)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from shared_simd_scan_amd import ScanEngine  # noqa: E402

eng = ScanEngine(0)
n, c = 1_000_000_000, 9
col = eng.generate("splitmix", n, c, 42)
bm = eng.alloc_bitmap(n)
h = torch.zeros(1, dtype=torch.int64, device="cuda")
mask = eng.scan_where("<", 256, col)[0]
keys = [(37 * k + 3) % 512 for k in range(40)]
for name, fn in (("scan_in P=40", lambda: eng.scan_in(keys, col, bitmap=bm, hits=h)),
                 ("scan_in P=40 AND mask", lambda: eng.scan_in(keys, col, and_mask=mask, bitmap=bm, hits=h))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    e1.synchronize()
    print(name, e0.elapsed_time(e1) / 20, "ms", int(h.item()), flush=True)

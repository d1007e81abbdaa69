#!/usr/bin/env python3
"""count-only scan (mi355_scan_combine_dev with bitmap_dev = NULL) against the bitmap-producing scan, and the fused
mask operations; 1e9 rows x 9 bit, launches back to back.  usage: python tools/count_only.py [rows] [bits]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from shared_simd_scan_amd import ScanEngine  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 9
eng = ScanEngine(0)
col = eng.generate("splitmix", n, c, 42)
bm = eng.alloc_bitmap(n)
mask = eng.scan_where("<", (1 << c) // 2, col)[0]
hits = torch.zeros(1, dtype=torch.int64, device="cuda")


def timed(name, fn, nbytes, burst=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(burst):
            fn()
        e1.record()
        e1.synchronize()
        best.append(e0.elapsed_time(e1) / burst)
    best.sort()
    print(f"{name:44s} median {best[2]:.4f} ms  best {best[0]:.4f} ms  {nbytes / best[2] / 1e6:8.1f} GB/s algorithmic  hits={int(hits.item())}", flush=True)


rd = n * c / 8
timed("scan_where == (bitmap + count)", lambda: eng.scan_where("==", 3, col, bitmap=bm, hits=hits), rd + n / 8)
timed("scan_combine == count only", lambda: eng.scan_combine("==", 3, col, hits=hits, count_only=True), rd)
timed("scan_combine between, count only", lambda: eng.scan_combine("between", 100, col, b=200, hits=hits, count_only=True), rd)
timed("scan_combine == AND mask (bitmap + count)", lambda: eng.scan_combine("==", 3, col, mask=mask, mask_op="and", bitmap=bm, hits=hits), rd + n / 4)
timed("scan_combine == OR mask (bitmap + count)", lambda: eng.scan_combine("==", 3, col, mask=mask, mask_op="or", bitmap=bm, hits=hits), rd + n / 4)
timed("scan_combine == AND mask, count only", lambda: eng.scan_combine("==", 3, col, mask=mask, mask_op="and", hits=hits, count_only=True), rd + n / 8)

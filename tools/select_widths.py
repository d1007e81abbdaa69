#!/usr/bin/env python3
"""tools/select_widths.py -- scan_select over the widths: both kernels (option select_kernel 1 / 2) at three selectivities.

    python tools/select_widths.py [rows] > profiles/rNN_select_widths.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from shared_simd_scan_amd import ScanEngine, lib  # noqa: E402
from shared_simd_scan_amd._capi import check  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
eng = ScanEngine(0)
ids = torch.zeros(n // 2 + n // 16 + 1024, dtype=torch.int64, device="cuda")
cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
print(f"{n} rows, random column; ms per launch (median of 7): select_kernel (one role) / select2_kernel (decoders + expanders)")
for c in (3, 5, 7, 9, 12, 13, 16, 17, 21, 25, 32):
    col = eng.generate("splitmix", n, c, 42)
    top = 1 << c
    cases = [("== 1", 0, 1), (f"< {max(1, top // 64)} (1/64)", 2, max(1, top // 64)), (f"< {top // 8} (1/8)", 2, top // 8), (f"< {top // 2} (1/2)", 2, top // 2)]
    for label, op, x in cases:
        res = []
        for kern in (1, 2):
            eng.set_option("select_kernel", kern)
            ts = []
            for rep in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib().mi355_scan_select_dev(eng._ctx, col.data.data_ptr(), n, c, op, x, 0, 0, None, 0, ids.data_ptr(), ids.numel() - 1024, cnt.data_ptr()))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res.append((sorted(ts[2:])[3], int(cnt[0].item())))
        assert res[0][1] == res[1][1], res
        print(f"c = {c:2d}  {label:22s} ids {res[0][1]:10d}   {res[0][0]:7.4f} / {res[1][0]:7.4f} ms   ratio {res[0][0] / res[1][0]:.2f}", flush=True)
    del col

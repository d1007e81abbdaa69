#!/usr/bin/env python3
"""tools/select_debug.py -- select2_kernel look-back diagnostics (option kernel_flags bit 14): polls, retries (a nearer chunk had
not published yet), LDS tag spins of the partner expanders, summed over the launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from shared_simd_scan_amd import ScanEngine, lib  # noqa: E402
from shared_simd_scan_amd._capi import check  # noqa: E402

eng = ScanEngine(0)
n, c = 1_000_000_000, 9
col = eng.generate("splitmix", n, c, 42)
eng.set_option("kernel_flags", 16384)
eng.set_option("select_kernel", 2)
for op, x, cap in ((0, 77, 4_000_000), (2, 8, 20_000_000), (2, 64, 130_000_000), (2, 256, 510_000_000)):  # MI355_CMP_EQ, MI355_CMP_LT
    ids = torch.zeros(cap + 1024, dtype=torch.int64, device="cuda")
    for rep in range(3):
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        check(lib().mi355_scan_select_dev(eng._ctx, col.data.data_ptr(), n, c, op, x, 0, 0, None, 0, ids.data_ptr(), cap, cnt.data_ptr()))
        torch.cuda.synchronize()
        v = cnt.cpu().tolist()
        st = ids[cap:cap + 512].cpu().numpy().reshape(64, 8)
        if rep == 2:
            t0 = st[0, 0]
            for g in range(20):
                r = (st[g] - t0) / 100.0  # 100 MHz ticks -> us
                print(f"   gen {g:2d}: start {r[0]:8.2f}  decoded {r[1]:8.2f}  look-back {r[2]:8.2f} .. {r[3]:8.2f}  expanders done {r[4]:8.2f}  barrier passed {r[5]:8.2f}")
        print(f"op {op} x {x}: count {v[0]}  polls {v[1]}  retries {v[2]}  partner tag spins {v[3]}  (chunks {(n + 65535) // 65536})", flush=True)

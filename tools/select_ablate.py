#!/usr/bin/env python3
"""tools/select_ablate.py -- select_kernel same-process A/B: how the chunks are handed out (tickets + barrier = the
product; by block index as in round 2, with / without the per-generation barrier) and the two timing ablations
(no expansion, no look-back: wrong ids by construction).  1e9 x 9 bit, selectivities 1/512 and 1/2.
option kernel_flags bits 9-12 = 512 no expansion, 1024 no look-back, 2048 chunks by block index, 4096 no barrier (with 2048)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from shared_simd_scan_amd import ScanEngine  # noqa: E402

eng = ScanEngine(0)
n, c = 1_000_000_000, 9
col = eng.generate("splitmix", n, c, 42)
hits = torch.zeros(1, dtype=torch.int64, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


print("count-only scan", timed(lambda: eng.scan_combine("==", 77, col, hits=hits, count_only=True)))
for sel, op, x, cap in (("1/512", "==", 77, 4_000_000), ("1/2", "<", 256, 520_000_000)):
    ids = torch.empty(cap, dtype=torch.int64, device="cuda")
    for rnd in range(2):
        for flags, name in ((0, "tickets + barrier (product)"), (2048, "by block index + barrier"), (2048 + 4096, "by block index, no barrier (round 2)"),
                            (512, "no expand"), (1024, "no look-back"), (1536, "decode + park only")):
            eng.set_option("kernel_flags", flags)
            print(f"select {sel:6s} {name:40s} {timed(lambda: eng.scan_select(op, x, col, capacity=cap), 10):8.4f} ms", flush=True)
    del ids
eng.set_option("kernel_flags", 0)

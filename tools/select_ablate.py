#!/usr/bin/env python3
"""tools/select_ablate.py -- select_kernel same-process A/B: how the chunks are handed out (tickets + barrier = the
product; by block index as in round 2, with / without the per-generation barrier) and the two timing ablations
(no expansion, no look-back: wrong ids by construction).  1e9 x 9 bit, selectivities 1/512 and 1/2.
option kernel_flags bits 9-13 = 512 no expansion, 1024 no look-back, 2048 chunks by block index, 4096 no barrier (with 2048),
8192 select_kernel (round 2's single-role kernel) instead of select2_kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from shared_simd_scan_amd import ScanEngine  # noqa: E402

eng = ScanEngine(0)
eng.set_option("select_kernel", 2)  # the flags below pick the kernel (8192 = select_kernel)
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
for sel, op, x, cap in (("1/512", "==", 77, 4_000_000), ("1/64", "<", 8, 17_000_000), ("1/8", "<", 64, 130_000_000), ("1/2", "<", 256, 520_000_000)):
    ids = torch.empty(cap, dtype=torch.int64, device="cuda")
    for rnd in range(1):
        for flags, name in ((0, "select2: decoders + expanders (product)"), (512, "select2, no expand"), (1024, "select2, no look-back"),
                            (1536, "select2, decode + park only"), (8192, "select_kernel: tickets + barrier"),
                            (8192 + 2048, "select_kernel: by block index + barrier"), (8192 + 2048 + 4096, "select_kernel: by block index, no barrier (round 2)"),
                            (8192 + 512, "select_kernel, no expand"), (8192 + 1024, "select_kernel, no look-back")):
            eng.set_option("kernel_flags", flags)
            print(f"select {sel:6s} {name:40s} {timed(lambda: eng.scan_select(op, x, col, capacity=cap), 10):8.4f} ms", flush=True)
    del ids
eng.set_option("kernel_flags", 0)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shared_simd_scan_amd import ScanEngine
eng = ScanEngine(0)
n, c = 1_000_000_000, 9
col = eng.generate("splitmix", n, c, 42)
hits = torch.zeros(1, dtype=torch.int64, device="cuda")
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
print("count-only scan", timed(lambda: eng.scan_combine("==", 77, col, hits=hits, count_only=True)))
for flags, name in ((0, "full"), (512, "no expand"), (1024, "no look-back"), (1536, "decode + park only")):
    eng.set_option("kernel_flags", flags)
    print(f"select {name:20s}", timed(lambda: eng.scan_select("==", 77, col, capacity=4_000_000)), flush=True)

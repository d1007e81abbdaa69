#!/usr/bin/env python3
"""tools/sweep.py -- width x blocks-per-CU x workload sweep on one GPU (tuning aid, not the bench).
usage: python tools/sweep.py [--rows N] [--bits 5,7,9,...] [--bpc 0,1,2,4] [--ops scan_eq,scan_range,shared_scan,decompress]
Prints one line per configuration: median / min ms over --reps launches (HIP events) and algorithmic GB/s."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--bits", default="5,7,9,12,17,21")
    ap.add_argument("--bpc", default="0,1,2,3,4")
    ap.add_argument("--ops", default="scan_eq,scan_range,shared_scan,decompress")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--aux", default="2")
    ap.add_argument("--nts", default="-1", help="scan_nt_stores values to sweep (-1 auto, 0 plain, 1 nt)")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import torch

    from shared_simd_scan_amd import ScanEngine

    eng = ScanEngine(0)
    n = args.rows
    nb = (n + 7) // 8
    rows = []
    for c in [int(x) for x in args.bits.split(",")]:
        col = eng.generate("splitmix", n, c, 42)
        bitmap = eng.alloc_bitmap(n)
        hits = torch.zeros(8, dtype=torch.int64, device="cuda")
        stride = (nb + 255) // 256 * 256
        out8 = torch.empty((8, stride), dtype=torch.uint8, device="cuda") if "shared_scan" in args.ops else None
        dec = torch.empty(n, dtype=torch.int32, device="cuda") if "decompress" in args.ops else None
        lo, hi = (1 << c) // 4, (1 << c) // 2
        keys = [(37 * k + 3) % (1 << c) for k in range(8)]
        steps = {
            "scan_eq": (lambda: eng.scan(keys[0], col, bitmap=bitmap, hits=hits[:1]), n * c / 8 + n / 8),
            "scan_range": (lambda: eng.scan_range(lo, hi, col, bitmap=bitmap, hits=hits[:1]), n * c / 8 + n / 8),
            "shared_scan": (lambda: eng.shared_scan(keys, col, out=out8, hits=hits), n * c / 8 + n),
            "decompress": (lambda: eng.decompress(col, out=dec), n * c / 8 + 4 * n),
        }
        for op in args.ops.split(","):
            fn, nbytes = steps[op]
            for aux in [int(x) for x in args.aux.split(",")]:
                eng.set_option("dma_aux", aux)
                for bpc, nts in [(int(x), int(y)) for x in args.bpc.split(",") for y in args.nts.split(",")]:
                    eng.set_option("max_blocks_per_cu", bpc)
                    eng.set_option("scan_nt_stores", nts)
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    ms = []
                    for _ in range(args.reps):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        fn()
                        e1.record()
                        e1.synchronize()
                        ms.append(e0.elapsed_time(e1))
                    ms.sort()
                    med, mn = ms[len(ms) // 2], ms[0]
                    row = {"op": op, "bits": c, "aux": aux, "bpc": bpc, "nts": nts, "med_ms": med, "min_ms": mn,
                           "gbs_med": nbytes / med / 1e6, "gbs_min": nbytes / mn / 1e6, "values_per_s": n / med * 1e3}
                    rows.append(row)
                    print(f"{op:12s} c={c:2d} aux={aux} bpc={bpc} nts={nts:2d}  med {med:8.4f} ms  min {mn:8.4f} ms  "
                          f"{row['gbs_med']:7.1f} GB/s  {row['values_per_s']:.3e} values/s", flush=True)
        del col, bitmap, out8, dec
        torch.cuda.empty_cache()
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()

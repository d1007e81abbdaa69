#!/usr/bin/env python3
"""tools/sweep_p.py -- shared scan over the number of predicates P (the reference's own sweep is P = 1..512,
scripts/prepare_shared_scan_results.py:28-31), both layouts, with and without hit counts.
usage: python tools/sweep_p.py [--rows N] [--bits C] [--P 1,2,4,8,...] [--reps R]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=250_000_000)
    ap.add_argument("--bits", type=int, default=9)
    ap.add_argument("--P", default="1,2,4,8,16,32,64,128,256,512")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--layouts", default="per_predicate,linear")
    ap.add_argument("--nts", default="-1")
    ap.add_argument("--bpc", default="0", help="max_blocks_per_cu values to sweep (0 = the engine's default)")
    ap.add_argument("--vpl", default="0", help="shared_vpl values to sweep (0 = the engine's choice, 64, 128)")
    ap.add_argument("--burst", type=int, default=1, help="launches per timed region (back to back: sustained rate)")
    ap.add_argument("--hits", default="1,0", help="1: with hit counts, 0: without")
    ap.add_argument("--unaligned", action="store_true", help="per-predicate bitmaps at a 16-byte-multiple stride (round 1) instead of whole 128-byte lines")
    args = ap.parse_args()
    import torch

    from shared_simd_scan_amd import ScanEngine

    eng = ScanEngine(0)
    n, c = args.rows, args.bits
    nb = (n + 7) // 8
    col = eng.generate("splitmix", n, c, 42)
    for P in [int(x) for x in args.P.split(",")]:
        keys = [(37 * k + 3) % (1 << c) for k in range(P)]
        for layout in args.layouts.split(","):
            if layout == "per_predicate":
                out = torch.empty((P, (nb + 255) // 256 * 256 if not args.unaligned else (nb + 15) // 16 * 16), dtype=torch.uint8, device="cuda")
            else:
                out = torch.empty(nb * P, dtype=torch.uint8, device="cuda")
            hits = torch.zeros(P, dtype=torch.int64, device="cuda")
            for nts, bpc, vpl in [(int(x), int(y), int(z)) for x in args.nts.split(",") for y in args.bpc.split(",")
                                  for z in args.vpl.split(",")]:
                eng.set_option("scan_nt_stores", nts)
                eng.set_option("max_blocks_per_cu", bpc)
                if args.vpl != "0":
                    eng.set_option("shared_vpl", vpl)
                for with_hits in [bool(int(h)) for h in args.hits.split(",")]:
                    fn = lambda: eng.shared_scan(keys, col, layout=layout, out=out, hits=hits if with_hits else False)  # noqa: E731
                    for _ in range(2):
                        fn()
                    torch.cuda.synchronize()
                    ms = []
                    for _ in range(args.reps):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(args.burst):
                            fn()
                        e1.record()
                        e1.synchronize()
                        ms.append(e0.elapsed_time(e1) / args.burst)
                    ms.sort()
                    med = ms[len(ms) // 2]
                    nbytes = n * c / 8 + n / 8 * P
                    print(f"P={P:4d} {layout:13s} nts={nts:2d} bpc={bpc} vpl={vpl:3d} hits={int(with_hits)}  med {med:8.4f} ms  {nbytes / med / 1e6:7.1f} GB/s  "
                          f"{n * P / med * 1e3:.3e} predicate-evals/s", flush=True)
            del out
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

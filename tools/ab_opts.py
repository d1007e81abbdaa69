#!/usr/bin/env python3
"""Same-process, interleaved A/B of engine options on one workload (MI355X guide rule 24: perf deltas come from
interleaved rounds in ONE process).

    python tools/ab_opts.py --workload scan_eq --bits 9 --opt scan_burst=0,1,2,4 [--rows 1e9] [--rounds 5] [--burst 50]
workloads: scan_eq | scan_range | scan_and (== with a fused AND mask) | count (count-only) | shared (needs --P, --layout)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from shared_simd_scan_amd import ScanEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="scan_eq")
ap.add_argument("--rows", type=float, default=1e9)
ap.add_argument("--bits", default="9")
ap.add_argument("--column", default="splitmix")
ap.add_argument("--opt", required=True, help="name=v1,v2,...")
ap.add_argument("--fixed", default="", help="name=value[,name=value] set once")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--burst", type=int, default=50)
ap.add_argument("--P", type=int, default=8)
ap.add_argument("--layout", default="per_predicate")
ap.add_argument("--hits", type=int, default=1)
args = ap.parse_args()

eng = ScanEngine(0)
name, vals = args.opt.split("=")
vals = [int(v) for v in vals.split(",")]
for kv in filter(None, args.fixed.split(",")):
    k, v = kv.split("=")
    eng.set_option(k, int(v))
n = int(args.rows)
for c in [int(x) for x in args.bits.split(",")]:
    col = eng.generate("splitmix", n, c, 42) if args.column == "splitmix" else eng.generate("mod", n, c, 5)
    nb = (n + 7) // 8
    hits = torch.zeros(max(args.P, 1), dtype=torch.int64, device="cuda")
    key = 3
    if args.workload == "shared":
        keys = [(37 * k + 3) % (1 << c) for k in range(args.P)]
        out = torch.empty((args.P, (nb + 255) // 256 * 256) if args.layout == "per_predicate" else (nb * args.P,), dtype=torch.uint8, device="cuda")
        fn = lambda: eng.shared_scan(keys, col, layout=args.layout, out=out, hits=hits if args.hits else False)  # noqa: E731
        nbytes = n * c / 8 + n / 8 * args.P
    else:
        bm = eng.alloc_bitmap(n)
        mask = eng.scan_where("<", (1 << c) // 2, col)[0]
        if args.workload == "scan_eq":
            fn = lambda: eng.scan(key, col, bitmap=bm, hits=hits)  # noqa: E731
            nbytes = n * c / 8 + n / 8
        elif args.workload == "scan_range":
            fn = lambda: eng.scan_range((1 << c) // 4, (1 << c) // 2, col, bitmap=bm, hits=hits)  # noqa: E731
            nbytes = n * c / 8 + n / 8
        elif args.workload == "scan_and":
            fn = lambda: eng.scan_combine("==", key, col, mask=mask, mask_op="and", bitmap=bm, hits=hits)  # noqa: E731
            nbytes = n * c / 8 + n / 4
        elif args.workload == "decompress":
            dec = torch.empty(n, dtype=torch.int32, device="cuda")
            fn = lambda: eng.decompress(col, out=dec)  # noqa: E731
            nbytes = n * c / 8 + 4 * n
        elif args.workload == "count":
            fn = lambda: eng.scan_combine("==", key, col, hits=hits, count_only=True)  # noqa: E731
            nbytes = n * c / 8
        else:
            raise SystemExit("unknown workload")
    times = {v: [] for v in vals}
    for rnd in range(args.rounds):
        for v in vals:
            eng.set_option(name, v)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.burst):
                fn()
            e1.record()
            e1.synchronize()
            times[v].append(e0.elapsed_time(e1) / args.burst)
    for v in vals:
        t = sorted(times[v])
        med = t[len(t) // 2]
        print(f"{args.workload:10s} c={c:2d} n={n:.0e} {name}={v:3d}  median {med:.4f} ms  best {t[0]:.4f}  {nbytes / med / 1e6:8.1f} GB/s algorithmic"
              f"  read {n * c / 8 / med / 1e6:8.1f} GB/s" + (f"  P={args.P} {args.layout} hits={args.hits}" if args.workload == "shared" else ""), flush=True)
    del col
    torch.cuda.empty_cache()

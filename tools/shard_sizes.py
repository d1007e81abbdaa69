#!/usr/bin/env python3
"""tools/shard_sizes.py -- one GPU, the shard sizes of the headline metric's strong scaling (ONE 1e9 x 9-bit
column in 1 / 2 / 4 / 8 row-range shards at 8192-row boundaries): equality scan + hit count, launches back to back.
What a shard of an N-GPU run costs on its own GPU, before a multi-GPU node exists: the launch-latency share at 8 shards,
and the scaling efficiency the scans alone allow (t(1 shard of 1e9) / (N x t(largest shard of N))).
usage: python tools/shard_sizes.py [--rows 1000000000] [--bits 9] [--steps 200]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--bits", type=int, default=9)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--worlds", default="1,2,4,8")
    args = ap.parse_args()
    import torch

    from shared_simd_scan_amd import ScanEngine
    from shared_simd_scan_amd.sharded import shard_rows

    eng = ScanEngine(0)
    c = args.bits
    base = None
    print(f"# {args.rows:.0e} x {c} bit, v = i % 5, key 3, scan + hit count, {args.steps} launches back to back per figure")
    print("# N  shard rows   kernel ms (HIP events)  wall ms/launch  values/s (shard)  TB/s (algorithmic)  strong-scaling efficiency of the scans")
    for world in [int(w) for w in args.worlds.split(",")]:
        ranges = shard_rows(args.rows, world)
        worst = None
        for r in sorted({0, world - 1}):  # a full-size shard and the ragged last one
            first, last = ranges[r]
            n = last - first
            col = eng.generate("mod", n, c, 5, first_row=first)
            bitmap, hits = eng.alloc_bitmap(n), torch.zeros(1, dtype=torch.int64, device="cuda")
            for _ in range(20):
                eng.scan(3, col, bitmap=bitmap, hits=hits)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(args.steps):
                eng.scan(3, col, bitmap=bitmap, hits=hits)
            e1.record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / args.steps * 1e3
            ms = e0.elapsed_time(e1) / args.steps
            assert int(hits.item()) == (last - 3 + 4) // 5 - (first - 3 + 4) // 5
            worst = max(worst or 0.0, wall)
            if world == 1:
                base = wall
            print(f"{world:3d} {n:11d}   {ms:10.4f}            {wall:10.4f}     {n / wall * 1e3:.3e}         {n * (c + 1) / 8 / ms / 1e9:6.3f}"
                  f"        {'' if r != world - 1 else f'{base / (world * worst):.3f}  (aggregate {args.rows / worst * 1e3:.3e} values/s)'}", flush=True)
            del col, bitmap
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

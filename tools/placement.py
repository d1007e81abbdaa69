#!/usr/bin/env python3
"""tools/placement.py -- does the PLACEMENT of the buffers in HBM change the scan / decompress time?
One process, one column content; the packed column and the outputs are carved out of one big pool at different
offsets (and from fresh allocations), each placement timed with HIP events over --reps back-to-back launches.
usage: python tools/placement.py [--rows N] [--bits C] [--reps R]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--bits", type=int, default=9)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--ops", default="scan_eq,decompress")
    args = ap.parse_args()
    import torch

    from shared_simd_scan_amd import ScanEngine
    from shared_simd_scan_amd.engine import PackedColumn, compressed_buffer_size, scan_output_buffer_size

    eng = ScanEngine(0)
    n, c = args.rows, args.bits
    src = eng.generate("splitmix", n, c, 42)
    pbytes = compressed_buffer_size(c, n)
    bbytes = scan_output_buffer_size(n)
    hits = torch.zeros(1, dtype=torch.int64, device="cuda")

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / args.reps

    MiB = 1 << 20
    pool = torch.empty(12 * 1024 * MiB, dtype=torch.uint8, device="cuda")
    print(f"pool at {pool.data_ptr():#x}", flush=True)
    ops = args.ops.split(",")
    # (column offset, output offset) inside the pool
    out_base = ((pbytes + 64 * MiB) // (2 * MiB) + 1) * 2 * MiB
    placements = [(0, out_base + d) for d in (0, 256, 4096, 64 * 1024, MiB, 2 * MiB + 4096, 3 * MiB, 16 * MiB + 65536, 128 * MiB,
                                               512 * MiB + 12288)]
    placements += [(d, out_base + 256 * MiB) for d in (256, 4096, 65536, MiB, 5 * MiB + 8192)]
    for coff, ooff in placements:
        colbuf = pool[coff:coff + pbytes]
        colbuf.copy_(src.data[:pbytes])
        col = PackedColumn(colbuf, n, c)
        line = f"col+{coff:#11x} out+{ooff:#11x}"
        if "scan_eq" in ops:
            bm = pool[ooff:ooff + bbytes]
            ms = timed(lambda: eng.scan(3, col, bitmap=bm, hits=hits))
            line += f"  scan_eq {ms:7.4f} ms {(n * c / 8 + n / 8) / ms / 1e6:7.0f} GB/s"
        if "decompress" in ops:
            dec = pool[ooff:ooff + 4 * n].view(torch.int32)
            ms = timed(lambda: eng.decompress(col, out=dec))
            line += f"  decompress {ms:7.4f} ms {(n * c / 8 + 4 * n) / ms / 1e6:7.0f} GB/s"
        print(line, flush=True)
    del pool
    torch.cuda.empty_cache()
    # fresh allocations, earlier ones kept alive so the addresses differ
    keep = []
    for i in range(6):
        colbuf = torch.empty(pbytes, dtype=torch.uint8, device="cuda")
        colbuf.copy_(src.data[:pbytes])
        col = PackedColumn(colbuf, n, c)
        bm = torch.empty(bbytes, dtype=torch.uint8, device="cuda")
        dec = torch.empty(n, dtype=torch.int32, device="cuda")
        line = f"fresh col {colbuf.data_ptr():#x} bm {bm.data_ptr():#x} dec {dec.data_ptr():#x}"
        if "scan_eq" in ops:
            ms = timed(lambda: eng.scan(3, col, bitmap=bm, hits=hits))
            line += f"  scan_eq {ms:7.4f} ms"
        if "decompress" in ops:
            ms = timed(lambda: eng.decompress(col, out=dec))
            line += f"  decompress {ms:7.4f} ms"
        print(line, flush=True)
        keep.append((colbuf, bm, dec))
        pad = torch.empty((i + 1) * 37 * MiB + 4096 * i, dtype=torch.uint8, device="cuda")
        keep.append(pad)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/bench_next.py -- the steps either side of the scan on one GPU, each against its
algorithmic HBM bytes: device packer / generators, bitmap combine / count / row-id materialisation, comparison
predicates with a fused AND-mask, IN-lists.  HIP events around --reps back-to-back launches.
usage: python tools/bench_next.py [--rows N] [--bits C] [--reps R]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def select_rows(args, eng, col, n, c, nb, pk, timed, report):
    """predicate -> row ids, fused (one launch) against the chain scan -> bitmap -> row ids (4 launches), per selectivity"""
    import torch

    key = 77
    out = torch.empty(nb + 64, dtype=torch.uint8, device="cuda")
    h1 = torch.zeros(1, dtype=torch.int64, device="cuda")
    sels = (("1/512", "==", key), ("1/64", "<", max(1, (1 << c) // 64)), ("1/8", "<", max(1, (1 << c) // 8)), ("1/2", "<", (1 << c) // 2))
    if args.only_select:
        sels = tuple(x for x in sels if x[0] in args.only_select.split(","))
    for name, op, x in sels:
        bm, hh = eng.scan_where(op, x, col)
        cnt = int(hh.item())
        ids = None

        def chain():
            nonlocal ids
            b2, _ = eng.scan_where(op, x, col, bitmap=out, hits=h1)
            ids = eng.bitmap_to_rowids(b2, n, capacity=cnt)

        def rowids_only():
            nonlocal ids
            ids = eng.bitmap_to_rowids(bm, n, capacity=cnt)

        def fused():
            nonlocal ids
            ids = eng.scan_select(op, x, col, capacity=cnt)

        report(f"bitmap_to_rowids selectivity {name}", timed(rowids_only, 5), nb + 8 * cnt, f"{cnt} ids")
        t_chain = timed(chain, 5)
        report(f"scan -> bitmap -> row ids, selectivity {name}", t_chain, pk + nb + 2 * nb + 8 * cnt, "4 launches; bitmap written once, read twice")
        t_fused = timed(fused, 5)
        report(f"scan_select (fused), selectivity {name}", t_fused, pk + 8 * cnt,
               f"1 launch, no bitmap in HBM: {3 * nb / 1e6:.0f} MB less traffic, {t_chain / t_fused:.2f}x the chain's speed")
        if name == "1/512":
            col_b = eng.generate("splitmix", n, 12, 4242)
            cnt_dev = torch.tensor([cnt], dtype=torch.int64, device="cuda")
            taken = torch.empty(cnt, dtype=torch.int32, device="cuda")
            report(f"gather: another column's values at those {cnt} ids", timed(lambda: eng.gather(col_b, ids[0][:cnt], cnt_dev, out=taken), 10),
                   12 * cnt, "(\"take\": two dwords per id; algorithmic = 8 B id + 4 B value per row)")
            del col_b, taken
        del ids
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--bits", type=int, default=9)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--select-only", dest="only_select", default="", help="only the selection rows, e.g. 1/2,1/8,1/64,1/512")
    args = ap.parse_args()
    import torch

    from shared_simd_scan_amd import ScanEngine

    eng = ScanEngine(0)
    n, c = args.rows, args.bits
    nb = (n + 7) // 8
    pk = n * c / 8

    def timed(fn, reps=args.reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def report(name, ms, nbytes, note=""):
        print(f"{name:44s} {ms:8.4f} ms  {nbytes / ms / 1e6:7.0f} GB/s algorithmic  {n / ms * 1e3:.3e} rows/s  {note}", flush=True)

    col = eng.generate("splitmix", n, c, 42)
    if args.only_select:
        return select_rows(args, eng, col, n, c, nb, pk, timed, report)
    report("generate splitmix (write packed)", timed(lambda: eng.generate("splitmix", n, c, 42), 5), pk)
    vals = eng.decompress(col)  # int32[n]: input of the device packer
    report("compress u32 -> packed", timed(lambda: eng.compress(vals, c), 5), 4 * n + pk)
    del vals
    torch.cuda.empty_cache()

    key = 77
    bm_a, hits = eng.scan_where("<", (1 << c) // 2, col)
    bm_b, _ = eng.scan_where(">=", (1 << c) // 8, col)
    out = torch.empty_like(bm_a)
    h1 = torch.zeros(1, dtype=torch.int64, device="cuda")
    for op in ("<", "!=", "between", "not_between"):
        report(f"scan_where {op}", timed(lambda: eng.scan_where(op, (1 << c) // 4, col, b=(1 << c) // 2, bitmap=out, hits=h1)), pk + nb)
    report("scan_where < AND mask (fused conjunction)",
           timed(lambda: eng.scan_where("<", (1 << c) // 4, col, and_mask=bm_b, bitmap=out, hits=h1)), pk + 2 * nb)
    for P in (4, 40, 400):
        keys = [(37 * k + 3) % (1 << c) for k in range(P)]
        report(f"scan_in P={P}", timed(lambda: eng.scan_in(keys, col, bitmap=out, hits=h1)), pk + nb)
    report("aggregate: sum / count / min / max of the column", timed(lambda: eng.aggregate(col)), pk, "one pass, nothing decoded to memory")
    report("aggregate under a bitmap (WHERE earlier predicate)", timed(lambda: eng.aggregate(col, mask=bm_a)), pk + nb)
    if c <= 14:
        report("histogram: rows per value (GROUP BY)", timed(lambda: eng.histogram(col)), pk, "2^c counters in LDS, one LDS atomic per value")
        report("histogram under a bitmap", timed(lambda: eng.histogram(col, mask=bm_a)), pk + nb)
    report("bitmap_combine AND (+popcount)", timed(lambda: eng.bitmap_combine("and", bm_a, bm_b, n, out=out)), 3 * nb)
    report("bitmap_count", timed(lambda: eng.bitmap_count(bm_a, n)), nb)
    # fused consumers: what leaves out the bitmap round trip through HBM
    report("scan == (bitmap + count)", timed(lambda: eng.scan(key, col, bitmap=out, hits=h1)), pk + nb)
    report("scan_combine == count only (no bitmap)", timed(lambda: eng.scan_combine("==", key, col, hits=h1, count_only=True)), pk,
           f"saves {nb / 1e6:.0f} MB of bitmap writes")
    report("scan_combine < OR mask (fused disjunction)",
           timed(lambda: eng.scan_combine("<", (1 << c) // 4, col, mask=bm_b, mask_op="or", bitmap=out, hits=h1)), pk + 2 * nb)
    report("scan_combine < AND mask, count only", timed(lambda: eng.scan_combine("<", (1 << c) // 4, col, mask=bm_b, hits=h1, count_only=True)),
           pk + nb)
    # two columns: one launch against scan + fused-mask scan
    col2 = eng.generate("splitmix", n, c, 4242)

    def two_launches():
        b1, _ = eng.scan_where("<", (1 << c) // 4, col, bitmap=out, hits=h1)
        eng.scan_combine(">=", (1 << c) // 8, col2, mask=b1, mask_op="and", bitmap=out, hits=h1)

    t2 = timed(two_launches)
    report("col1 < a AND col2 >= b: scan + fused-mask scan", t2, 2 * pk + 3 * nb, f"2 launches, {nb / 1e6:.0f} MB bitmap written + read in between")
    t1 = timed(lambda: eng.scan2(col, "<", (1 << c) // 4, col2, ">=", (1 << c) // 8, combine="and", bitmap=out, hits=h1))
    report("col1 < a AND col2 >= b: mi355_scan2_dev", t1, 2 * pk + nb, f"1 launch, no intermediate bitmap: {t2 / t1:.2f}x")
    report("col1 < a AND col2 >= b: scan2, count only", timed(lambda: eng.scan2(col, "<", (1 << c) // 4, col2, ">=", (1 << c) // 8, hits=h1, count_only=True)), 2 * pk)
    del col2
    torch.cuda.empty_cache()
    select_rows(args, eng, col, n, c, nb, pk, timed, report)


if __name__ == "__main__":
    main()

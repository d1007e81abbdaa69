#!/usr/bin/env python3
"""tools/cli_sweep.py -- the reference's shared-scan sweep (scripts/prepare_shared_scan_results.py:22-31: data size 40 MB,
1 repetition, predicate_count = 1 .. 512 in steps of one, `<binary> 40 1 sharedscan P`, stdout parsed with the rule of its
parse_output, :14-20) driven against cli/shared_simd_scan_mi355.  Writes the same CSV (data_size, predicate_count, variant,
avg_runtime_ms) and fails when any run prints a self-check line ("first mismatch at index ...") or exits non-zero.
usage: python tools/cli_sweep.py [--binary cli/shared_simd_scan_mi355] [--first 1] [--last 512] [--step 1] [--reps 1] [--out FILE]"""
import argparse
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_output(output):
    """scripts/prepare_shared_scan_results.py:14-20, restated: lines that start with '*' are `* <variant>: <avg> ms; [...]`"""
    rows = []
    for line in output.splitlines():
        if not line.startswith("*"):
            continue
        variant = line[2:].split(": ")[0]
        avg_runtime_ms = line.split(": ")[1].split("; ")[0][:-2]
        rows.append((variant, avg_runtime_ms))
    return rows


def sweep(binary, data_size=40, reps=1, counts=range(1, 513), out=sys.stdout, extra=()):
    w = csv.writer(out)
    w.writerow(["data_size", "predicate_count", "variant", "avg_runtime_ms"])
    bad = []
    for p in counts:
        res = subprocess.run([binary, str(data_size), str(reps), "sharedscan", str(p), *extra], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True)
        rows = parse_output(res.stdout)
        if res.returncode != 0 or "mismatch" in res.stdout or len(rows) < 2:
            bad.append((p, res.returncode, [l for l in res.stdout.splitlines() if "mismatch" in l][:2], res.stderr[-300:]))
        for variant, ms in rows:
            float(ms)  # must be a number, as the reference's plot script reads it (scripts/plot_shared_scan_results.py)
            w.writerow([data_size, p, variant, ms])
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--binary", default=os.path.join(ROOT, "cli", "shared_simd_scan_mi355"))
    ap.add_argument("--first", type=int, default=1)
    ap.add_argument("--last", type=int, default=512)
    ap.add_argument("--step", type=int, default=1)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--data-size", type=int, default=40)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    out = open(args.out, "w", newline="") if args.out else sys.stdout
    bad = sweep(args.binary, args.data_size, args.reps, range(args.first, args.last + 1, args.step), out)
    if args.out:
        out.close()
    for b in bad:
        print("FAILED P=%d rc=%d %s %s" % b, file=sys.stderr)
    print(f"# {len(range(args.first, args.last + 1, args.step))} runs, {len(bad)} with a mismatch line or a non-zero status", file=sys.stderr)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/profile_shared.py -- rocprofv3 evidence for shared-scan kernels at given widths / key counts (run through gpurun).

For every (bits, P) asked for: tools/profile_cmd.py's passes (one `--kernel-trace --stats` pass, separate `--pmc` passes)
over `python3 tools/sweep_p.py --bits C --P P ...`, one summary block per configuration with the algorithmic TB/s.

usage: python tools/profile_shared.py --bits 17,21,25 --P 16,64 [--layout per_predicate] [--hits 1] [--rows 250000000] --out FILE"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import profile_cmd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", default="17,21,25")
    ap.add_argument("--P", default="16,64")
    ap.add_argument("--layout", default="per_predicate")
    ap.add_argument("--hits", default="1")
    ap.add_argument("--rows", type=int, default=250_000_000)
    ap.add_argument("--out", required=True)
    ap.add_argument("--scratch", default=os.path.join(ROOT, "gpurun_out", "prof_shared"))
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="MI355_KERNEL_FLAGS for the profiled process (A/B switches)")
    args = ap.parse_args()
    if args.flags:
        os.environ["MI355_KERNEL_FLAGS"] = str(args.flags)
    lines = [f"# tools/profile_shared.py: shared scan, {args.rows} rows, layout {args.layout}, hit counts {args.hits}, kernel_flags {args.flags};",
             "# rocprofv3 --kernel-trace --stats + separate --pmc passes over `python3 tools/sweep_p.py`; TB/s = (n c / 8 + n P / 8) / average",
             "# kernel time; counters are sums over all waves of one launch (mean over the launches of the pass)"]
    for c in [int(x) for x in args.bits.split(",")]:
        for P in [int(x) for x in args.P.split(",")]:
            tag = f"c{c}_P{P}_{args.layout}_h{args.hits}_f{args.flags}"
            work = ["tools/sweep_p.py", "--rows", str(args.rows), "--bits", str(c), "--P", str(P), "--layouts", args.layout, "--hits", args.hits]
            res = profile_cmd.profile(work, "shared_", os.path.join(args.scratch, tag), pmc=not args.no_pmc, stats_extra=("--reps", "20"),
                                      pmc_extra=("--reps", "3"))
            lines.append("")
            lines.append(f"c = {c}, P = {P}: {res.get('name', '')}")
            if "avg_us" in res:
                nbytes = args.rows * c / 8 + args.rows / 8 * P
                lines.append(f"    => {nbytes / (res['avg_us'] * 1e-6) / 1e12:5.2f} TB/s algorithmic, {args.rows * P / (res['avg_us'] * 1e-6):.3e} predicate evaluations/s")
            lines += profile_cmd.describe(res)
            with open(args.out, "w") as f:
                f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/profile_cmd.py -- rocprofv3 evidence for ONE kernel of an arbitrary python command (run through gpurun).

    python tools/profile_cmd.py --match select_kernel --out gpurun_out/x.txt [--label "..."] -- tools/bench_next.py --select-only 1/2

One `--kernel-trace --stats` pass (calls, average duration of the kernel whose name contains --match and has the largest
total time) and separate `--pmc` passes (never combined with other tracing domains), then a summary block: the kernel,
its average launch time, VGPRs / LDS / scratch of the dispatch, and what the waves did.  This process never touches the
GPU itself; rocprofv3 is started with the python interpreter directly behind `--`."""
import argparse
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY",
    "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS",
    "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU",
]


def newest(pattern):
    return sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)[-1:]


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        return subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, env=env, cwd=ROOT).returncode


def profile(work, match, base, pmc=True, stats_extra=(), pmc_extra=()):
    """work: the python command line behind `python3`; -> dict(name, calls, avg_us, min_us, max_us, meta, counters) or None"""
    os.makedirs(base, exist_ok=True)
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(base, "stats"), "--", "python3"]
             + list(work) + list(stats_extra), os.path.join(base, "stats.log"))
    best = None
    for f in newest(os.path.join(base, "stats", "**", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if match in r["Name"] and (best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"])):
                best = r
    if best is None:
        return {"error": f"no kernel matching {match!r} in the trace (rc {rc})"}
    res = {"name": best["Name"], "calls": int(best["Calls"]), "avg_us": float(best["AverageNs"]) / 1e3,
           "min_us": float(best["MinNs"]) / 1e3, "max_us": float(best["MaxNs"]) / 1e3, "meta": {}, "counters": {}}
    if not pmc:
        return res
    counters = collections.defaultdict(list)
    for gi, grp in enumerate(GROUPS):
        d = os.path.join(base, f"pmc{gi}")
        run(["rocprofv3", "--pmc"] + grp.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3"] + list(work)
            + list(pmc_extra), os.path.join(base, f"pmc{gi}.log"))
        for f in newest(os.path.join(d, "**", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"] == res["name"]:
                    counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    res["meta"] = {k: r.get(k) for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                                         "Accum_VGPR_Count", "SGPR_Count")}
    res["counters"] = {k: sum(v) / len(v) for k, v in counters.items()}
    return res


def describe(res):
    """summary lines for one profile() result"""
    if "error" in res:
        return ["    " + res["error"]]
    lines = [f"    calls {res['calls']}  avg {res['avg_us']:9.2f} us  min {res['min_us']:9.2f}  max {res['max_us']:9.2f}"]
    meta, m = res["meta"], res["counters"]
    if meta:
        lines.append(f"    dispatch: grid {meta.get('Grid_Size')} x {meta.get('Workgroup_Size')}, VGPRs {meta.get('VGPR_Count')} (+{meta.get('Accum_VGPR_Count')} acc), "
                     f"SGPRs {meta.get('SGPR_Count')}, LDS {meta.get('LDS_Block_Size')} B / block, scratch {meta.get('Scratch_Size')} B")
    if m.get("SQ_BUSY_CYCLES"):
        busy = m["SQ_BUSY_CYCLES"]
        wc = m.get("SQ_WAVE_CYCLES", 0.0)

        def pct(x, of):
            return f"{100.0 * m.get(x, 0.0) / of:5.1f} %" if of else "  n/a"

        lines.append(f"    waves {m.get('SQ_WAVES', 0):.0f}; instructions: VALU {m.get('SQ_INSTS_VALU', 0) / 1e6:8.1f} M, LDS {m.get('SQ_INSTS_LDS', 0) / 1e6:7.1f} M, "
                     f"SALU {m.get('SQ_INSTS_SALU', 0) / 1e6:7.1f} M, VMEM rd {m.get('SQ_INSTS_VMEM_RD', 0) / 1e6:6.2f} M, wr {m.get('SQ_INSTS_VMEM_WR', 0) / 1e6:6.2f} M")
        lines.append(f"    of the waves' cycles: waiting (any) {pct('SQ_WAIT_ANY', wc)}, waiting to issue {pct('SQ_WAIT_INST_ANY', wc)}, "
                     f"waiting for LDS issue {pct('SQ_WAIT_INST_LDS', wc)}, issuing {pct('SQ_ACTIVE_INST_ANY', wc)}")
        lines.append(f"    of the SIMD-cycles (4 x busy cycles): VALU busy {pct('SQ_ACTIVE_INST_VALU', busy * 4)}, LDS busy {pct('SQ_ACTIVE_INST_LDS', busy * 4)}, "
                     f"scalar busy {pct('SQ_ACTIVE_INST_SCA', busy * 4)}; LDS bank-conflict cycles / LDS active cycles "
                     f"{pct('SQ_LDS_BANK_CONFLICT', m.get('SQ_LDS_IDX_ACTIVE', 0.0))}")
        lines.append(f"    cycles spent issuing VMEM writes {m.get('SQ_INST_CYCLES_VMEM_WR', 0) / 1e6:8.1f} M, reads {m.get('SQ_INST_CYCLES_VMEM_RD', 0) / 1e6:8.1f} M "
                     f"(wave cycles {wc / 1e6:9.1f} M)")
        if m.get("SQ_LEVEL_WAVES") and m.get("GRBM_GUI_ACTIVE"):
            lines.append(f"    resident waves (SQ_LEVEL_WAVES / GRBM_GUI_ACTIVE, whole chip): {m['SQ_LEVEL_WAVES'] / m['GRBM_GUI_ACTIVE']:.0f}")
    return lines


def main():
    if "--" not in sys.argv:
        print(__doc__)
        sys.exit(2)
    cut = sys.argv.index("--")
    ap = argparse.ArgumentParser()
    ap.add_argument("--match", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--label", default="")
    ap.add_argument("--scratch", default=os.path.join(ROOT, "gpurun_out", "prof_cmd"))
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--append", action="store_true")
    args = ap.parse_args(sys.argv[1:cut])
    work = sys.argv[cut + 1:]
    tag = "".join(ch if ch.isalnum() else "_" for ch in (args.label or " ".join(work)))[:80]
    res = profile(work, args.match, os.path.join(args.scratch, tag), pmc=not args.no_pmc)
    lines = [f"{args.label or ' '.join(work)}: {res.get('name', '')}", f"    command: python3 {' '.join(work)}"] + describe(res)
    with open(args.out, "a" if args.append else "w") as f:
        f.write("\n".join(lines) + "\n\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

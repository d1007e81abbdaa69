#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory into profiles/<tag>_<workload>.{md,json} and update
profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact
for 16-B-per-lane streaming stores.  Each counter group was collected in its own rocprofv3 --pmc pass."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _sha():
    from shared_simd_scan_amd.build import kernel_sources_sha

    return kernel_sources_sha()


def main():
    src, workload, tag, rows, bits = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    kmatch = {"scan_eq": "scan_burst_kernel<%d, 0," % bits, "scan_range": "scan_burst_kernel<%d, 1," % bits,
              "shared_scan": "shared_lut_kernel<%d," % bits, "decompress": "decompress_kernel<%d," % bits}[workload]
    stats = None
    newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1:]  # noqa: E731  (re-runs leave older files)
    for f in newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if kmatch in r["Name"]:
                stats = r
    counters = collections.defaultdict(list)
    meta = {}
    pmc_files = []
    for d in glob.glob(os.path.join(src, "pmc_*")):
        if os.path.isdir(d):
            pmc_files += newest(os.path.join(d, "*", "*_counter_collection.csv"))
    for f in pmc_files:
        for r in csv.DictReader(open(f)):
            if kmatch in r["Kernel_Name"]:
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                          "SGPR_Count")}
    mean = {k: sum(v) / len(v) for k, v in counters.items()}
    out = {"workload": workload, "rows": rows, "bits": bits, "kernel": stats["Name"] if stats else None,
           "kernel_calls": int(stats["Calls"]) if stats else None,
           "kernel_avg_ns": float(stats["AverageNs"]) if stats else None,
           "kernel_min_ns": float(stats["MinNs"]) if stats else None,
           "kernel_max_ns": float(stats["MaxNs"]) if stats else None, "dispatch": meta, "counters_mean": mean}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        rd = mean["FETCH_SIZE"] * 1024 * 2  # gfx950: FETCH_SIZE counts 128-B requests at 64 B
        wr = mean["WRITE_SIZE"] * 1024
        out["hbm_read_bytes_per_launch"] = rd
        out["hbm_write_bytes_per_launch"] = wr
        out["hbm_bytes_per_launch"] = rd + wr
    if "GRBM_GUI_ACTIVE" in mean and stats:
        out["effective_clock_ghz"] = mean["GRBM_GUI_ACTIVE"] / 8 / float(stats["AverageNs"])
    # the L2's memory-side interface (TCC -> Infinity Fabric / HBM): Little's law on the request-level counters
    mem = {}
    if mean.get("TCC_EA0_RDREQ_sum") and "TCC_EA0_RDREQ_LEVEL_sum" in mean:
        mem["read_latency_tcc_cycles"] = mean["TCC_EA0_RDREQ_LEVEL_sum"] / mean["TCC_EA0_RDREQ_sum"]
        if mean.get("TCC_CYCLE_sum"):
            mem["reads_in_flight_per_channel"] = mean["TCC_EA0_RDREQ_LEVEL_sum"] / mean["TCC_CYCLE_sum"]
            mem["read_dram_credit_stall_frac"] = mean.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", 0.0) / mean["TCC_CYCLE_sum"]
    if mean.get("TCC_EA0_WRREQ_sum") and "TCC_EA0_WRREQ_LEVEL_sum" in mean:
        mem["write_latency_tcc_cycles"] = mean["TCC_EA0_WRREQ_LEVEL_sum"] / mean["TCC_EA0_WRREQ_sum"]
        mem["write_stall_per_request"] = mean.get("TCC_EA0_WRREQ_STALL_sum", 0.0) / mean["TCC_EA0_WRREQ_sum"]
        mem["write_dram_credit_stall_per_request"] = mean.get("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", 0.0) / mean["TCC_EA0_WRREQ_sum"]
    if mem:
        out["memory_side"] = mem
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    base = os.path.join(ROOT, "profiles", f"{tag}_{workload}_{rows:.0e}x{bits}".replace("+", ""))
    json.dump(out, open(base + ".json", "w"), indent=1)
    with open(base + ".md", "w") as f:
        f.write(f"# rocprofv3 summary: {workload}, {rows:.0e} rows x {bits} bit, 1x MI355X ({tag})\n\n")
        f.write("Command: `bash tools/profile.sh %s %s` (pass 1 `rocprofv3 --kernel-trace --stats`, then one `--pmc` "
                "pass per counter group), on `python3 bench.py --workload %s --no-cpu-baseline`.\n\n" % (workload, tag, workload))
        if stats:
            f.write("## kernel-trace --stats\n\n| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n")
            f.write(f"| `{stats['Name']}` | {stats['Calls']} | {float(stats['AverageNs']):.0f} | {stats['MinNs']} | {stats['MaxNs']} |\n\n")
        f.write(f"dispatch: {meta}\n\n## PMC (mean per launch of the kernel)\n\n| counter | value |\n|---|---|\n")
        for k in sorted(mean):
            f.write(f"| {k} | {mean[k]:.1f} |\n")
        if "hbm_bytes_per_launch" in out:
            f.write("\n## HBM traffic per launch (gfx950-corrected)\n\n")
            f.write(f"- read  = FETCH_SIZE x 1024 x 2 = {out['hbm_read_bytes_per_launch']:.4e} B\n")
            f.write(f"- write = WRITE_SIZE x 1024     = {out['hbm_write_bytes_per_launch']:.4e} B\n")
            f.write(f"- total = {out['hbm_bytes_per_launch']:.4e} B\n")
        if mem:
            f.write("\n## L2 <-> memory interface (TCC_EA0_*; level / requests = average time a request is outstanding)\n\n")
            for k, v in mem.items():
                f.write(f"- {k} = {v:.3f}\n")
        if "effective_clock_ghz" in out:
            f.write(f"\neffective clock = GRBM_GUI_ACTIVE / 8 / kernel time = {out['effective_clock_ghz']:.2f} GHz\n")
    if "hbm_bytes_per_launch" in out:
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            table = json.load(open(tpath))
        except (OSError, ValueError):
            table = {}
        table[f"{workload}:{rows}:{bits}"] = {"hbm_bytes_per_launch": out["hbm_bytes_per_launch"],
                                              "read": out["hbm_read_bytes_per_launch"],
                                              "write": out["hbm_write_bytes_per_launch"], "source": os.path.basename(base) + ".json",
                                              "kernel_sources_sha": _sha()}
        json.dump(table, open(tpath, "w"), indent=1)
    # the JSON line bench.py printed in pass 1 (its roofline.kernel_ms comes from HIP events in the SAME run as the
    # --stats average above) and the raw --stats kernel table
    log = os.path.join(src, "stats.log")
    if os.path.exists(log):
        lines = [ln for ln in open(log) if ln.startswith("{") and '"roofline"' in ln]
        if lines:
            bench = json.loads(lines[-1])
            json.dump(bench, open(os.path.join(ROOT, "profiles", f"{tag}_{workload}_bench_under_rocprof.json"), "w"), indent=1)
            if stats:
                with open(base + ".md", "a") as f:
                    f.write(f"\nbench.py in the same run: HIP-event kernel time {bench['roofline']['kernel_ms'] * 1e6:.0f} ns per launch, "
                            f"wall {bench['ms_per_step'] * 1e6:.0f} ns per step (rocprofv3 average above: {float(stats['AverageNs']):.0f} ns)\n")
    for f in newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
        open(os.path.join(ROOT, "profiles", f"{tag}_{workload}_kernel_stats.csv"), "w").write(open(f).read())
    print(open(base + ".md").read())


if __name__ == "__main__":
    main()

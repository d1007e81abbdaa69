#!/usr/bin/env python3
"""tools/doc_tables.py -- the measured tables of DESIGN.md / README.md, generated from the tracked files under profiles/.

    python tools/doc_tables.py [--tag r03]            prints every block
    python tools/doc_tables.py --write                rewrites the blocks between the markers
                                                      <!-- BEGIN GENERATED: <name> --> ... <!-- END GENERATED: <name> -->
                                                      in DESIGN.md and README.md

No figure in a generated block exists anywhere but in the named profiles/ file; tests/test_docs_follow_profiles.py
fails when a block in the documents differs from what this script makes of the committed files."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _read(name):
    with open(os.path.join(PROF, name)) as f:
        return f.read()


def _sweep_p(name):
    """tools/sweep_p.py lines -> {(P, layout, hits): (ms, GB/s)}"""
    out = {}
    for line in _read(name).splitlines():
        m = re.match(r"P=\s*(\d+) (\S+)\s+nts=.*hits=(\d)\s+med\s+([\d.]+) ms\s+([\d.]+) GB/s", line)
        if m:
            out[(int(m.group(1)), m.group(2), int(m.group(3)))] = (float(m.group(4)), float(m.group(5)))
    return out


def block_headline(tag):
    b = json.load(open(os.path.join(PROF, f"{tag}_bench_line.json")))
    r, cb = b["roofline"], b["cpu_baseline"]
    rows = ["| figure | value | source |", "|---|---|---|",
            f"| `value` (scanned values/s, 1 GPU, {b['steps']} steps) | {b['value']:.4g} | `profiles/{tag}_bench_line.json` |",
            f"| `ms_per_step` (wall) / `roofline.kernel_ms` (HIP events) | {b['ms_per_step']:.4f} / {r['kernel_ms']:.4f} ms | same |",
            f"| `roofline.achieved` = 1.25e9 B / kernel time | {r['achieved']:.0f} GB/s = {r['frac']:.3f} of 8000 | same |",
            f"| read stream (`read_gb_per_s`, `read_frac`) | {r['read_gb_per_s']:.0f} GB/s = {r['read_frac']:.3f} | same |",
            f"| `cpu_baseline` ({cb['kind']}, {cb['cores']} thread, {cb.get('host_cpu', '')}) | {cb['value']:.3g} values/s ({cb['ms']:.1f} ms), bitmap equals GPU: {cb['bitmap_equals_gpu']} | same |"]
    if "all_cores" in cb:
        rows.append(f"| the same reference function over {cb['all_cores']['cores']} host threads | {cb['all_cores']['value']:.3g} values/s ({cb['all_cores']['ms']:.2f} ms) | same |")
    return "\n".join(rows)


def block_rocprof(tag):
    rows = ["| workload | kernel | calls | rocprofv3 avg µs | bench.py HIP-event µs (same run) | algorithmic B / launch | TB/s | frac of 8 TB/s | PMC HBM bytes / algorithmic |",
            "|---|---|---|---|---|---|---|---|---|"]
    algo = {"scan_eq": 1.25e9, "scan_range": 1.25e9, "shared_scan": 2.125e9, "decompress": 5.125e9}
    for w in ("scan_eq", "scan_range", "shared_scan", "decompress"):
        d = json.load(open(os.path.join(PROF, f"{tag}_{w}_1e09x9.json")))
        bj = json.load(open(os.path.join(PROF, f"{tag}_{w}_bench_under_rocprof.json")))
        us = d["kernel_avg_ns"] / 1e3
        tb = algo[w] / (us * 1e-6) / 1e12
        name = re.sub(r"^void mi355::", "", d["kernel"]).replace("(mi355::ScanArgs)", "").replace("(mi355::DecompArgs)", "")
        rows.append(f"| {w} | `{name}` | {d['kernel_calls']} | {us:.2f} | {bj['roofline']['kernel_ms'] * 1e3:.2f} | {algo[w]:.4g} | {tb:.2f} | {tb / 8:.3f} | "
                    f"{d['hbm_bytes_per_launch'] / algo[w]:.4f} |")
    rows.append("")
    rows.append(f"(`profiles/{tag}_<workload>_1e09x9.json`, `{tag}_<workload>_bench_under_rocprof.json`; `tools/profile.sh <workload> {tag}`)")
    return "\n".join(rows)


def block_shared_all_p(tag):
    d = _sweep_p(f"{tag}_shared_scan_all_P.txt")
    ps = sorted({k[0] for k in d})

    def rng(layout, lo, hi):
        v = [d[(p, layout, 1)][1] / 1000 for p in ps if lo <= p <= hi and (p, layout, 1) in d]
        return f"{min(v):.2f}–{max(v):.2f}" if v else "—"

    rows = ["| key counts P (every P in the range, with hit counts) | per-predicate TB/s | linear TB/s |", "|---|---|---|"]
    for lo, hi in ((1, 8), (9, 16), (17, 32), (33, 48), (49, 64)):
        rows.append(f"| {lo} … {hi} | {rng('per_predicate', lo, hi)} | {rng('linear', lo, hi)} |")
    big = [p for p in ps if p > 64]
    for p in big:
        a, b = d.get((p, "per_predicate", 1)), d.get((p, "linear", 1))
        rows.append(f"| {p} | {a[1] / 1000:.2f} | {b[1] / 1000:.2f} |")
    rows.append("")
    rows.append(f"(2.5e8 rows × 9 bit, random column, keys (37k + 3) mod 512, one box; `profiles/{tag}_shared_scan_all_P.txt`, `tools/sweep_p.py`)")
    return "\n".join(rows)


def block_shared_p_sweep(tag):
    d = _sweep_p(f"{tag}_shared_scan_P_sweep.txt")
    ps = sorted({k[0] for k in d})
    rows = ["| P | per-predicate, with / without hit counts (TB/s) | linear, with / without (TB/s) |", "|---|---|---|"]
    for p in ps:
        def f(layout):
            a, b = d.get((p, layout, 1)), d.get((p, layout, 0))
            return f"{a[1] / 1000:.2f} / {b[1] / 1000:.2f}" if a and b else "—"
        rows.append(f"| {p} | {f('per_predicate')} | {f('linear')} |")
    rows.append("")
    rows.append(f"(2.5e8 rows × 9 bit, launches back to back; `profiles/{tag}_shared_scan_P_sweep.txt`)")
    return "\n".join(rows)


def block_wide_widths(tag):
    def parse(name):
        out = {}
        cur = None
        for line in _read(name).splitlines():
            m = re.match(r"c = (\d+), P = (\d+): void mi355::(\w+)<", line)
            if m:
                cur = (int(m.group(1)), int(m.group(2)))
                out[cur] = {"kernel": m.group(3)}
            m = re.match(r"\s+=>\s+([\d.]+) TB/s", line)
            if m and cur:
                out[cur]["tbs"] = float(m.group(1))
            m = re.match(r"\s+calls \d+\s+avg\s+([\d.]+) us", line)
            if m and cur:
                out[cur]["us"] = float(m.group(1))
            m = re.match(r"\s+dispatch: .*VGPRs (\d+)", line)
            if m and cur:
                out[cur]["vgpr"] = int(m.group(1))
        return out

    before, after = parse(f"{tag}_wide_widths_before.txt"), parse(f"{tag}_wide_widths_after.txt")
    rows = ["| c | P | round 2 kernel: avg µs, TB/s | round 3 kernel: avg µs, TB/s |", "|---|---|---|---|"]
    for key in sorted(before):
        b, a = before[key], after.get(key, {})
        rows.append(f"| {key[0]} | {key[1]} | `{b['kernel']}` {b['us']:.0f} µs, {b['tbs']:.2f} | `{a.get('kernel', '?')}` {a.get('us', 0):.0f} µs, {a.get('tbs', 0):.2f} |")
    rows.append("")
    rows.append(f"(per-predicate bitmaps with hit counts, 2.5e8 rows, rocprofv3 `--kernel-trace --stats` averages of 22 launches; `profiles/{tag}_wide_widths_before.txt`, "
                f"`{tag}_wide_widths_after.txt`: dispatch sizes and PMC counters are in the files)")
    return "\n".join(rows)


def block_select(tag):
    rows = ["| selectivity | scan → bitmap → row ids (4 launches) | `scan_select` (1 launch) | fused ÷ chain speed |", "|---|---|---|---|"]
    chain, fused = {}, {}
    for line in _read(f"{tag}_next_rows_1e9x9.txt").splitlines():
        m = re.match(r"scan -> bitmap -> row ids, selectivity (\S+)\s+([\d.]+) ms", line)
        if m:
            chain[m.group(1)] = float(m.group(2))
        m = re.match(r"scan_select \(fused\), selectivity (\S+)\s+([\d.]+) ms", line)
        if m:
            fused[m.group(1)] = float(m.group(2))
    for s in chain:
        rows.append(f"| {s} | {chain[s]:.4f} ms | {fused[s]:.4f} ms | {chain[s] / fused[s]:.2f} × |")
    rows.append("")
    rows.append(f"(1e9 rows × 9 bit, `select2_kernel`; `profiles/{tag}_next_rows_1e9x9.txt`; both kernels at every selectivity, same process: "
                f"`profiles/{tag}_select_ab.txt`)")
    return "\n".join(rows)


def block_shards(tag):
    rows = ["| GPUs N | rows of a shard | kernel µs (HIP events) | wall µs per launch | shard values/s | efficiency of the scans alone |", "|---|---|---|---|---|---|"]
    for line in _read(f"{tag}_shard_sizes_1gpu.txt").splitlines():
        m = re.match(r"\s*(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.e+]+)\s+([\d.]+)\s*([\d.]*)", line)
        if m:
            rows.append(f"| {m.group(1)} | {m.group(2)} | {float(m.group(3)) * 1e3:.1f} | {float(m.group(4)) * 1e3:.1f} | {m.group(5)} | {m.group(7) or ''} |")
    rows.append("")
    rows.append(f"(ONE GPU running a shard of ONE 1e9 × 9-bit column split N ways at 8192-row boundaries, scan + hit count, 200 launches back to back; "
                f"efficiency = t(N = 1) ÷ (N × t(slowest shard)); `profiles/{tag}_shard_sizes_1gpu.txt`, `tools/shard_sizes.py`)")
    return "\n".join(rows)


BLOCKS = {"headline": block_headline, "rocprof": block_rocprof, "shared_all_p": block_shared_all_p, "shared_p_sweep": block_shared_p_sweep,
          "wide_widths": block_wide_widths, "select": block_select, "shards": block_shards}


def render(tag):
    return {name: fn(tag) for name, fn in BLOCKS.items()}


def splice(text, blocks):
    for name, body in blocks.items():
        pat = re.compile(r"(<!-- BEGIN GENERATED: %s -->\n)(.*?)(<!-- END GENERATED: %s -->)" % (name, name), re.S)
        text = pat.sub(lambda m: m.group(1) + body + "\n" + m.group(3), text)
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    blocks = render(args.tag)
    if args.write:
        for doc in ("DESIGN.md", "README.md"):
            p = os.path.join(ROOT, doc)
            old = open(p).read()
            new = splice(old, blocks)
            if new != old:
                open(p, "w").write(new)
                print("updated", doc)
    else:
        for name, body in blocks.items():
            print(f"<!-- BEGIN GENERATED: {name} -->\n{body}\n<!-- END GENERATED: {name} -->\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark: equality scan of a 1e9-value 9-bit packed column per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (one scan_eq launch: bitmap + hit count) over the rank's
row-range shard, the column already resident in HBM; no data-path collective inside the timed region.
`value` is the headline of SURVEY 8e: ONE column of --rows rows (default 1e9) split over the N ranks at
8192-row boundaries ("scaling": "strong").  The weak figure -- every rank its own --rows rows -- is
measured in the same run and printed beside it (`weak_values_per_s`); at N = 1 they are one measurement.
The RCCL bitmap gather is timed separately after the scans and reported as `gather_ms`; an exchange that
fails or never returns still lets the scan line through, and then the run exits NON-ZERO (4 / 3).

Prints ONE JSON line on rank 0 (see DESIGN.md "measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 is what a copy achieves


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--tune", action="store_true", help="let the engine measure 1 / 2 / 4 resident blocks per CU on this device first "
                    "(mi355_tune_dev, untimed setup); off by default so that every launch of the kernel in a rocprofv3 trace of this "
                    "command is a launch of the timed configuration")
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="values per GPU")
    ap.add_argument("--bits", type=int, default=9)
    ap.add_argument("--column", choices=["mod5", "random"], default="mod5",
                    help="mod5: v=i%%5, key 3 (src/benchmark.cpp:173,:150); random: splitmix64(42,i)&mask, key v[12345]")
    ap.add_argument("--workload", choices=["scan_eq", "scan_range", "shared_scan", "decompress"], default="scan_eq")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--layout", choices=["per_predicate", "linear"], default="per_predicate",
                    help="shared_scan output: one bitmap per key, or the reference's linear layout (byte of group g, key k at g*8+k)")
    ap.add_argument("--pipelined-gather", action="store_true",
                    help="N>1: also time ShardedColumn.scan_pipelined (chunked scan, asynchronous gathers overlapping it)")
    ap.add_argument("--store-policy", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="result stores of the scans: -1 engine default, 0 plain, 1 non-temporal, 2 write-through (tuning)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="which partition is `value` (both are measured and printed at N > 1; they coincide at N = 1): strong "
                         "(default) = ONE column of --rows rows, row-range sharded at 8192-row boundaries (SURVEY 8e: the headline "
                         "metric at 2 / 4 / 8 GPUs); weak = --rows per GPU")
    ap.add_argument("--gather-timeout", type=float, default=180.0, help="N>1: seconds the exchange step may take before it is given up")
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the multi-core CPU leg (0 = every CPU this process may use: affinity mask / cgroup quota)")
    return ap.parse_args()


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota when one is set"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(args, col, gpu_bitmap, key):
    """The reference's CPU path on this host's cores, on the SAME column (downloaded from HBM), never a scalar straw man:
    kind 'reference'        : oracle/_ref, the reference's own scan_256_unrolled (its fastest single-thread variant,
                              src/simd_scan.cpp:273), when the prebuilt .so travelled here;
    kind 'avx2-restatement' : oracle/oracle_avx2.c, the own AVX2 restatement of that function (pinned to the reference's
                              golden vectors, tests/test_oracle_golden.py) -- what a clean checkout has.
    Both legs are reported when both exist.  `cores` = threads actually used: 1 for the headline figure (the reference is
    single-threaded), and an all-cores leg (row-range slices over the CPUs this process may use) beside it."""
    import numpy as np

    from oracle import RefLib, oracle, ref_available

    n, c = col.n, col.c
    packed = col.data.cpu().numpy()
    nb = (n + 7) // 8
    gpu_host = gpu_bitmap[:nb].cpu().numpy()
    O = oracle()
    share = host_cpu_share()
    threads = max(1, min(args.cpu_threads if args.cpu_threads > 0 else share, share))
    host = {"host_cpu": _cpu_model(), "host_cores": os.cpu_count(), "host_cpu_share": share}
    reps = max(1, args.cpu_reps)

    restated = None
    if O.avx2_available() and c <= 25:
        out, hits, secs = O.scan_eq_avx2(packed, n, c, key, threads=1, reps=reps)
        t = float(np.median(secs))
        restated = {"value": n / t, "unit": "values/s", "cores": 1, "kind": "avx2-restatement",
                    "sample": f"full column ({n} values), oracle_avx2.c (AVX2 in-place compare, 32 values per bitmap word), "
                              f"median of {reps} reps, 1 thread",
                    "ms": t * 1e3, "gb_per_s": n * c / 8 / t / 1e9, "bitmap_equals_gpu": bool(np.array_equal(out, gpu_host)),
                    "hits": hits}
        if threads > 1:
            out, hits_mt, secs = O.scan_eq_avx2(packed, n, c, key, threads=threads, reps=max(3, reps))
            best = float(min(secs))
            restated["all_cores"] = {"value": n / best, "unit": "values/s", "cores": threads, "ms": best * 1e3,
                                     "hits": hits_mt, "bitmap_equals_gpu": bool(np.array_equal(out, gpu_host)),
                                     "sample": f"same column, OpenMP row ranges over {threads} threads, best of {max(3, reps)}"}
    if c == 9 and ref_available(9):
        R = RefLib(9)
        secs, out, hits = R.scan_timed("scan_256_unrolled", key, packed, n, reps)
        t = float(np.median(secs))
        res = {"value": n / t, "unit": "values/s", "cores": 1, "kind": "reference",
               "sample": f"full column ({n} values), scan_256_unrolled, median of {reps} reps, 1 thread",
               "ms": t * 1e3, "gb_per_s": n * c / 8 / t / 1e9, "bitmap_equals_gpu": bool(np.array_equal(out[:nb], gpu_host))}
        # the same reference function on row-range slices, one thread per slice (the reference itself is
        # single-threaded; this is what its AVX2 path gives when the host's cores share the column)
        if threads > 1:
            from concurrent.futures import ThreadPoolExecutor

            per = -(-n // threads)
            per = -(-per // 8192) * 8192
            slices = [(a, min(n, a + per)) for a in range(0, n, per)]

            def run(sl):
                a, b = sl
                return R.scan_timed("scan_256_unrolled", key, packed[a * c // 8:], b - a, 1)[2]

            best = None
            with ThreadPoolExecutor(len(slices)) as ex:
                for _ in range(3):
                    t0 = time.perf_counter()
                    hits_mt = sum(ex.map(run, slices))
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
            res["all_cores"] = {"value": n / best, "unit": "values/s", "cores": len(slices), "ms": best * 1e3,
                                "hits": int(hits_mt), "sample": "same column, row-range slices, one reference call per thread"}
        if restated is not None:
            res["avx2_restatement"] = restated  # the clean-checkout leg, for comparison on the same host
        res.update(host)
        return res
    if restated is not None:
        restated.update(host)
        return restated
    raise RuntimeError("cpu_baseline: neither oracle/_ref nor an AVX2 host: refusing to time a scalar straw man "
                       "(run with --no-cpu-baseline)")


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run around it: this process becomes the
    launcher.  It touches neither torch.cuda nor HIP; it starts N fresh rank processes (one per GPU) through
    torch.distributed.run, lets rank 0's JSON line through on stdout and exits with the children's status."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU fallback"
    # BENCH_BACKEND=gloo rehearses the N>1 code path on a box with fewer GPUs than ranks (ranks share devices);
    # the judged runs use nccl (= RCCL over xGMI), one rank per GPU
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from shared_simd_scan_amd import ScanEngine, kernel_name
    from shared_simd_scan_amd.sharded import shard_rows

    eng = ScanEngine(local_rank)
    if args.store_policy >= 0:
        eng.set_option("scan_nt_stores", args.store_policy)
    c = args.bits
    mask = (1 << c) - 1
    keys8 = list(range(8))
    key = 3
    if args.column == "random":
        # key = v[12345] of the global column, read back through the engine itself
        key = int(eng.decompress(eng.generate("splitmix", 1, c, 42, first_row=12345)).cpu()[0].item()) & mask

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(x):
        if world == 1:
            return [float(x)]
        out = [None] * world
        dist.all_gather_object(out, float(x))
        return [float(v) for v in out]

    def run_phase(scaling):
        """one timed measurement: this rank's row range of the `scaling` partition, resident in HBM, W warm-up steps, then
        exactly K steps between barrier + synchronize on both sides; wall time = MAX over ranks"""
        if scaling == "strong":  # ONE column of --rows rows split over the ranks at 8192-row boundaries
            first, last = shard_rows(args.rows, world)[rank]
            n, total_rows = last - first, args.rows
        else:                    # every rank owns --rows rows: rank r holds global rows [r*rows, (r+1)*rows)
            first, n, total_rows = rank * args.rows, args.rows, world * args.rows
        if args.column == "mod5":
            # eq / range / decompress: v = i % 5, key 3 (src/benchmark.cpp:173,:150); shared scan: v = i % 8, keys 0..7
            # (src/benchmark.cpp:277,:205-209 -- BASELINE config 4)
            col = eng.generate("mod", n, c, 8 if args.workload == "shared_scan" else 5, first_row=first)
        else:
            col = eng.generate("splitmix", n, c, 42, first_row=first)
        nb = (n + 7) // 8
        if args.workload == "scan_eq":
            bitmap, hits = eng.alloc_bitmap(n), torch.zeros(1, dtype=torch.int64, device="cuda")
            step = lambda: eng.scan(key, col, bitmap=bitmap, hits=hits)  # noqa: E731
            algo_bytes = n * c / 8 + n / 8
            kname = kernel_name("scan_eq", c)
        elif args.workload == "scan_range":
            bitmap, hits = eng.alloc_bitmap(n), torch.zeros(1, dtype=torch.int64, device="cuda")
            lo, hi = (1 << c) // 4, (1 << c) // 2
            step = lambda: eng.scan_range(lo, hi, col, bitmap=bitmap, hits=hits)  # noqa: E731
            algo_bytes = n * c / 8 + n / 8
            kname = kernel_name("scan_range", c)
        elif args.workload == "shared_scan":
            stride = (nb + 255) // 256 * 256  # per-predicate bitmaps start on whole 128-byte lines (mi355_bitmap_stride)
            bitmap = torch.empty((8, stride) if args.layout == "per_predicate" else (nb * 8,), dtype=torch.uint8, device="cuda")
            hits = torch.zeros(8, dtype=torch.int64, device="cuda")
            step = lambda: eng.shared_scan(keys8, col, layout=args.layout, out=bitmap, hits=hits)  # noqa: E731
            algo_bytes = n * c / 8 + 8 * n / 8
            kname = kernel_name("shared_scan", c)
        else:
            out = torch.empty(n, dtype=torch.int32, device="cuda")
            bitmap, hits = out, None
            step = lambda: eng.decompress(col, out=out)  # noqa: E731
            algo_bytes = n * c / 8 + 4 * n
            kname = kernel_name("decompress", c)

        # --tune: load-time tuning (mi355_tune_dev), what a service does once after loading a column -- the engine measures
        # 1 / 2 / 4 resident blocks per CU on THIS device and keeps the fastest; setup, outside the warm-up and the timed region
        tuned = {}
        if args.tune and args.workload != "shared_scan":
            tuned = eng.tune(col, "decompress" if args.workload == "decompress" else "scan")

        for _ in range(args.warmup):
            step()
        sync_all()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            step()
        ev1.record()
        sync_all()
        elapsed = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the stream the kernels ran on
        per_rank_ms = gather_floats(dev_ms)
        if world > 1:
            elapsed = reduce_max(elapsed)

        # correctness of what was timed
        expect_hits = None
        if args.workload == "scan_eq" and args.column == "mod5":
            lo_r, hi_r = first, first + n
            expect_hits = (hi_r - 3 + 4) // 5 - (lo_r - 3 + 4) // 5  # rows i in [lo,hi) with i%5==3
            got = int(hits.item())
            assert got == expect_hits, f"hits {got} != {expect_hits}"
        if args.workload == "shared_scan" and args.column == "mod5":
            # rows i in [first, first+n) with i % 8 == k
            want = [(first + n - 1 - k) // 8 - (first - 1 - k) // 8 for k in range(8)]
            got = [int(x) for x in hits.tolist()]
            assert got == want, f"shared-scan hits {got} != {want}"
        return {"scaling": scaling, "first": first, "n": n, "nb": nb, "total_rows": total_rows, "col": col, "bitmap": bitmap,
                "hits": hits, "elapsed": elapsed, "dev_ms": dev_ms, "per_rank_ms": per_rank_ms, "algo_bytes": algo_bytes,
                "kname": kname, "tuned": tuned, "expect_hits": expect_hits,
                "value": total_rows / (elapsed / args.steps)}

    # The headline (SURVEY 8e / BASELINE.json "1e9 x 9-bit column, 1/2/4/8 GPU") is ONE column of --rows rows in N
    # row-range shards: strong scaling.  The weak figure (every rank its own --rows rows) is measured in the same run and
    # printed beside it; at N = 1 the two partitions are the same measurement.  --scaling weak swaps which one is `value`.
    phases = {args.scaling: run_phase(args.scaling)}
    other = "weak" if args.scaling == "strong" else "strong"
    if world > 1:
        phases[other] = run_phase(other)
    else:
        phases[other] = phases[args.scaling]
    ph = phases[args.scaling]
    n, nb, first, total_rows = ph["n"], ph["nb"], ph["first"], ph["total_rows"]
    col, bitmap, hits, expect_hits = ph["col"], ph["bitmap"], ph["hits"], ph["expect_hits"]
    elapsed, dev_ms, algo_bytes, kname, tuned = ph["elapsed"], ph["dev_ms"], ph["algo_bytes"], ph["kname"], ph["tuned"]
    ranges = shard_rows(args.rows, world) if args.scaling == "strong" else [(r * args.rows, (r + 1) * args.rows) for r in range(world)]

    xr = {"gather_ms": None, "gather_error": None, "pipelined_ms": None, "gather_via": None, "done": False, "ranks_seen": None,
          "comm_world": None}
    force = os.environ.get("BENCH_FORCE_EXCHANGE_FAILURE", "")  # rehearsal of the failure paths: raise | hang | hang<rank>

    def exchange_phase():
        """the exchange step, timed apart from the scan; runs in a helper thread so that a bootstrap or collective that
        never returns (a mis-configured fabric) is SEEN: the scan line is still printed, then the rank exits non-zero"""
        torch.cuda.set_device(local_rank)
        if world > 1 and not args.no_gather and args.workload in ("scan_eq", "scan_range"):
            # final exchange step of the north star: per-shard bitmaps -> rank 0 over xGMI, through the C ABI's RCCL entry
            # points (mi355_comm_create / mi355_gather_bitmaps_dev / mi355_allreduce_hits_dev); a gloo group (rehearsal on a
            # box with fewer GPUs than ranks) uses the torch.distributed transport instead
            from shared_simd_scan_amd.sharded import ExchangeUnavailable, ShardedColumn, TorchExchange, make_exchange

            if force == "hang" or force == f"hang{rank}":
                time.sleep(10 ** 6)
            sizes = [(b - a + 7) // 8 for a, b in ranges]
            offs = [sum(sizes[:r]) for r in range(world)]
            ex = None
            try:
                ex = make_exchange(eng, None)  # collective: the same transport, or ExchangeUnavailable, on EVERY rank
            except ExchangeUnavailable as e:  # RCCL bootstrap through the C ABI failed everywhere: say so, use torch's group
                xr["gather_error"] = f"C-ABI RCCL exchange unavailable ({e}); torch.distributed used"
                ex = TorchExchange(None)
            xr["gather_via"] = ex.name
            try:
                if force == "raise":
                    raise RuntimeError("forced by BENCH_FORCE_EXCHANGE_FAILURE")
                xr["comm_world"] = ex.info()[0]  # mi355_comm_info: the communicator's own idea of its size
                ones = torch.ones(1, dtype=torch.int64, device="cuda")
                xr["ranks_seen"] = int(ex.sum_hits(ones, engine=eng).item())  # ranks that actually took part in an all-reduce
                full = ex.gather(bitmap[:nb], sizes, dst=0, engine=eng)  # warm
                total_hits = ex.sum_hits(hits, engine=eng)
                sync_all()
                g0 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    full = ex.gather(bitmap[:nb], sizes, dst=0, out=full, engine=eng)
                sync_all()
                xr["gather_ms"] = reduce_max((time.perf_counter() - g0) / reps * 1e3)
                if expect_hits is not None:
                    lo_r, hi_r = ranges[0][0], ranges[-1][1]
                    want = (hi_r - 3 + 4) // 5 - (lo_r - 3 + 4) // 5
                    assert int(total_hits.item()) == want, f"all-reduced hits {int(total_hits.item())} != {want}"
                if rank == 0:  # every slice must have arrived where the packed layout puts it
                    assert torch.equal(full[:nb], bitmap[:nb]), "gathered bitmap: the root's own slice differs"
                    if expect_hits is not None:
                        # i % 5 column (period 40 rows = 5 bitmap bytes), root's slice starts at row 0: the slice of a rank
                        # that starts at row a (a multiple of 8) holds the root's bytes from byte (a / 8) % 5 on
                        for r in range(1, world):
                            shift = (ranges[r][0] // 8) % 5
                            m = min(sizes[r] - 1, nb - shift - 1)
                            if m > 0:
                                assert torch.equal(full[offs[r]: offs[r] + m], bitmap[shift: shift + m]), \
                                    f"gathered bitmap: slice of rank {r} differs"
            except Exception as e:  # the scan figures above stand on their own; report the exchange step as failed
                xr["gather_error"] = f"{type(e).__name__}: {e}"
                xr["failed"] = True
            if args.pipelined_gather and args.workload == "scan_eq" and xr["gather_ms"] is not None:
                try:
                    sc = ShardedColumn(total_rows, c, engine=eng, exchange=ex)
                    sc.ranges = list(ranges)
                    sc.first, sc.last, sc.rows = first, first + n, n
                    sc.col = col
                    sc.scan_pipelined(key, dst=0, chunks=4)  # warm
                    sync_all()
                    g0 = time.perf_counter()
                    for _ in range(5):
                        full_p, hits_p = sc.scan_pipelined(key, dst=0, chunks=4)
                    sync_all()
                    xr["pipelined_ms"] = reduce_max((time.perf_counter() - g0) / 5 * 1e3)
                    if rank == 0:
                        assert torch.equal(full_p, full), "pipelined gather: bitmap differs from the plain gather"
                except Exception as e:
                    xr["gather_error"] = f"pipelined: {type(e).__name__}: {e}"
                    xr["failed"] = True

        xr["done"] = True

    import threading

    th = threading.Thread(target=exchange_phase, daemon=True)
    th.start()
    th.join(args.gather_timeout)
    timed_out = th.is_alive()
    if timed_out:
        xr["gather_error"] = f"exchange step did not finish within {args.gather_timeout} s (scan figures unaffected); exit status 3"
    gather_ms, gather_error, pipelined_ms, gather_via = xr["gather_ms"], xr["gather_error"], xr["pipelined_ms"], xr["gather_via"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = ph["value"]
        achieved = algo_bytes / (dev_ms * 1e-3) / 1e9
        read_gbs = n * c / 8 / (dev_ms * 1e-3) / 1e9
        traffic, traffic_src = _pmc_traffic(args, kname, n)
        headline = args.workload == "scan_eq" and c == 9 and total_rows == 1_000_000_000 * (world if args.scaling == "weak" else 1)
        colname = "i%8" if args.workload == "shared_scan" and args.column == "mod5" else args.column
        what = (f"ONE {args.rows:.0e}x{c}bit column in {world} row-range shard(s) of <= {max(b - a for a, b in ranges)} rows"
                if args.scaling == "strong" else f"{args.rows:.0e}x{c}bit per GPU")
        line = {
            # BASELINE.json: "scanned values/sec + achieved HBM GB/s, 1e9 x 9-bit column, 1/2/4/8 GPU";
            # `value` is the values/sec half, `achieved_hbm_gb_per_s` (= roofline.achieved) the other
            "metric": "scanned values/sec + achieved HBM GB/s, 1e9 x 9-bit column (equality scan, bitmap + hit count out)"
            if headline else f"{args.workload} values/sec, {args.rows:.0e} x {c}-bit column",
            "value": value, "unit": "values/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32" if args.workload != "decompress" else "i32", "data": "synthetic",
            "config": {"workload": f"{args.workload} {what}, column={colname}, "
                                   + (f"keys=0..7, layout={args.layout}" if args.workload == "shared_scan" else f"key={key}"),
                       "rows_total": total_rows, "rows_per_gpu": [b - a for a, b in ranges], "bits": c,
                       "parallelism": f"row-range shards x{world}", "tuned_blocks_per_cu": tuned or None},
            # both partitions, measured in this one run (identical at N = 1): the driver's scaling curve can use either
            "strong_values_per_s": phases["strong"]["value"], "weak_values_per_s": phases["weak"]["value"],
            "strong_ms_per_step": phases["strong"]["elapsed"] / args.steps * 1e3,
            "weak_ms_per_step": phases["weak"]["elapsed"] / args.steps * 1e3,
            "per_rank_kernel_ms": ph["per_rank_ms"],
            "weak_per_rank_kernel_ms": phases["weak"]["per_rank_ms"], "strong_per_rank_kernel_ms": phases["strong"]["per_rank_ms"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "kernel_ms": dev_ms, "algorithmic_bytes": algo_bytes, "rows_per_launch": n,
                         # BASELINE.md section 3 words the target on the READ stream (>= 5.6 TB/s = 0.70 of peak)
                         "read_gb_per_s": read_gbs, "read_frac": read_gbs / HBM_PEAK_GBS},
            # all ranks together: algorithmic bytes of every shard over the slowest rank's kernel time
            "achieved_hbm_gb_per_s": sum((b - a) * (algo_bytes / max(n, 1)) for a, b in ranges) / (max(ph["per_rank_ms"]) * 1e-3) / 1e9,
            "hits": int(hits.sum().item()) if hits is not None else None,
        }
        if world > 1:
            line["ranks_seen"] = xr["ranks_seen"]
            line["comm_world"] = xr["comm_world"]
        if gather_via is not None:
            line["gather_via"] = gather_via
        if gather_ms is not None:
            line["gather_ms"] = gather_ms
            line["gather_gb_per_s"] = (total_rows / 8 - nb) / (gather_ms * 1e-3) / 1e9
            line["end_to_end_values_per_s"] = total_rows / ((ms_per_step + gather_ms) * 1e-3)
        if pipelined_ms is not None:  # one query = chunked scan with the gathers overlapping it (vs ms_per_step + gather_ms)
            line["pipelined_scan_gather_ms"] = pipelined_ms
        if gather_error is not None:
            line["gather_error"] = gather_error
        if world == 1 and not args.no_cpu_baseline and args.workload == "scan_eq":
            line["cpu_baseline"] = cpu_baseline(args, col, bitmap, key)
        print(json.dumps(line), flush=True)
    # A failed or stuck exchange is a FAILURE of the run: the scan line above is complete and printed, and the driver
    # sees a non-zero status (3 = a rank never came back from the exchange, 4 = the exchange raised).  Never status 0.
    if timed_out:
        sys.stdout.flush()
        sys.stderr.write(f"bench.py rank {rank}: {xr['gather_error']}\n")
        sys.stderr.flush()
        os._exit(3)  # the helper thread sits inside a collective: it cannot be joined, and neither can the process group
    if world > 1:
        dist.destroy_process_group()
    if xr.get("failed"):
        sys.stderr.write(f"bench.py rank {rank}: exchange failed: {xr['gather_error']}\n")
        sys.exit(4)


def _pmc_traffic(args, kname, n):
    """(HBM bytes per launch, provenance) from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: collected
    in separate --pmc passes and corrected as MI355X_MICROARCH.md prescribes, FETCH_SIZE x2 on gfx950).  The figure is
    only reported while the device-side sources still hash to what the profile was taken on: otherwise (None, why)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    ent = table.get(f"{args.workload}:{n}:{args.bits}")
    if not ent:
        return None, "no committed PMC profile for this workload / size"
    from shared_simd_scan_amd.build import kernel_sources_sha

    sha = kernel_sources_sha()
    if ent.get("kernel_sources_sha") != sha:
        return None, f"stale: profiles/{ent.get('source')} was taken on kernel sources {ent.get('kernel_sources_sha')}, now {sha}"
    return ent.get("hbm_bytes_per_launch"), f"profiles/{ent.get('source')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, kernel sources {sha})"


if __name__ == "__main__":
    main()

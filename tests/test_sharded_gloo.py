"""CPU: the multi-GPU (N>1) path on gloo, world_size 2.

The per-shard scan on a CPU rank is the ORACLE (injected here, in tests only: the product's ShardedColumn always
uses the HIP engine); what is under test is the product's sharding logic: row-range partition, bitmap gather,
hit-count reduction, ragged last shard.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_rows_partition():
    from shared_simd_scan_amd.sharded import SHARD_ALIGN, shard_rows

    for n in (0, 1, 8191, 8192, 8193, 100_003, 1_000_000_000, 8_000_000_000):
        for world in (1, 2, 3, 4, 8):
            r = shard_rows(n, world)
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
            for (a, b), (c, d) in zip(r, r[1:]):
                assert b == c and a <= b
            for a, b in r:
                assert a % SHARD_ALIGN == 0 or a == n  # every shard starts on a tile boundary
    # BASELINE config 5: 8e9 rows over 8 GPUs -> 1e9 rows each (1e9 is not a multiple of 8192: rounded up)
    r = shard_rows(8_000_000_000, 8)
    assert all(b - a <= 1_000_005_632 for a, b in r) and sum(b - a for a, b in r) == 8_000_000_000


class OracleEngine:
    """CPU stand-in for ScanEngine inside this test: same methods, backed by oracle.c."""

    def __init__(self):
        from oracle import oracle

        self.O = oracle()

    def generate(self, kind, n, c, param=0, first_row=0):
        vals = self.O.gen_values(kind, n, c, param, first=first_row)
        return (self.O.pack(vals, c), n, c)

    def slice_rows(self, col, first, last):
        packed, n, c = col
        assert first % 128 == 0 and 0 <= first <= last <= n
        return (packed[first * c // 8:], last - first, c)

    def scan(self, key, col):
        packed, n, c = col
        bm, hits = self.O.scan_eq(packed, n, c, key)
        return torch.from_numpy(bm), torch.tensor([hits], dtype=torch.int64)

    def scan_range(self, lo, hi, col):
        packed, n, c = col
        bm, hits = self.O.scan_range(packed, n, c, lo, hi)
        return torch.from_numpy(bm), torch.tensor([hits], dtype=torch.int64)


    _CMP = {"==": lambda v, a, b: v == a, "<": lambda v, a, b: v < a, "between": lambda v, a, b: (v >= a) & (v <= b)}

    def scan_select(self, op, a, col, capacity, b=0, first_row=0):
        """the fused selection, restated through the oracle's decompress (test stand-in only)"""
        packed, n, c = col
        v = self.O.decompress(packed, n, c).astype(np.int64) if n else np.zeros(0, dtype=np.int64)
        rows = np.nonzero(self._CMP[op](v, a, b))[0].astype(np.int64) + first_row
        ids = torch.zeros(max(capacity, 1), dtype=torch.int64)
        k = min(rows.shape[0], capacity)
        ids[:k] = torch.from_numpy(rows[:k])
        return ids, torch.tensor([rows.shape[0]], dtype=torch.int64)


    def aggregate(self, col, mask=None):
        packed, n, c = col
        v = self.O.decompress(packed, n, c).astype(np.uint64) if n else np.zeros(0, dtype=np.uint64)
        if mask is not None:
            bits = np.unpackbits(mask.numpy()[: (n + 7) // 8], bitorder="little")[:n].astype(bool)
            v = v[bits]
        if v.shape[0] == 0:
            return torch.tensor([0, 0, -1, 0], dtype=torch.int64)
        return torch.tensor([int(v.sum()), int(v.shape[0]), int(v.min()), int(v.max())], dtype=torch.int64)


def worker(rank, world, port, n, c, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from shared_simd_scan_amd.sharded import ShardedColumn

        O = oracle()
        sc = ShardedColumn(n, c, engine=OracleEngine())
        sc.generate("splitmix", 42)
        key = int(O.gen_values("splitmix", 1, c, 42, first=12345 % max(n, 1))[0])
        full, hits = sc.scan(key, dst=0)
        lo, hi = (1 << c) // 4, (1 << c) // 2
        full_r, hits_r = sc.scan_range(lo, hi, dst=0)
        full_p, hits_p = sc.scan_pipelined(key, dst=0, chunks=3)  # chunked scan + asynchronous gathers: same result
        ids, nids = sc.select("between", lo, dst=0, b=hi)           # global row ids on the root, the count everywhere
        local_bm, _ = sc.engine.scan_range(lo, hi, sc.col)
        agg = sc.aggregate(mask=local_bm)                           # aggregates over the predicate's rows, on every rank
        vals_all = O.gen_values("splitmix", n, c, 42).astype(np.uint64)
        sel_all = vals_all[(vals_all >= lo) & (vals_all <= hi)]
        assert agg == ((int(sel_all.sum()), int(sel_all.shape[0]), int(sel_all.min()), int(sel_all.max())) if sel_all.shape[0] else (0, 0, None, 0)), agg
        assert sc.aggregate()[:2] == (int(vals_all.sum()), n)
        if rank == 0:
            vals = O.gen_values("splitmix", n, c, 42)
            want_ids = np.nonzero((vals >= lo) & (vals <= hi))[0].astype(np.int64)
            assert nids == want_ids.shape[0] and np.array_equal(ids.numpy(), want_ids)
            packed = O.pack(vals, c)
            ref, ref_hits = O.scan_eq(packed, n, c, key)
            ref_r, ref_hits_r = O.scan_range(packed, n, c, lo, hi)
            ok = (np.array_equal(full.numpy(), ref) and int(hits.item()) == ref_hits
                  and np.array_equal(full_p.numpy(), ref) and int(hits_p.item()) == ref_hits
                  and np.array_equal(full_r.numpy(), ref_r) and int(hits_r.item()) == ref_hits_r)
            q.put(("ok" if ok else "mismatch", sc.ranges))
        else:
            assert full is None and ids is None
            assert int(hits.item()) >= 0  # every rank gets the column-wide count
            assert nids == int(hits_r.item())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [100_003, 16_384, 8192 * 3 + 5, 5, 8192 * 9 + 1, 8192 * 8])
def test_sharded_scan_world2_gloo(n):
    world, c = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, n, c, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    status, ranges = q.get(timeout=10)
    assert status == "ok", ranges


def test_c_abi_shard_rows_matches_python():
    import ctypes as C

    from shared_simd_scan_amd import lib
    from shared_simd_scan_amd.sharded import shard_rows

    L = lib()
    for n in (0, 5, 8192, 100_003, 1_000_000_000, 8_000_000_000):
        for world in (1, 2, 3, 8):
            py = shard_rows(n, world)
            for r in range(world):
                a, cnt = C.c_uint64(), C.c_uint64()
                assert L.mi355_shard_rows(n, world, r, C.byref(a), C.byref(cnt)) == 0
                assert (a.value, a.value + cnt.value) == py[r]
    a, cnt = C.c_uint64(), C.c_uint64()
    assert L.mi355_shard_rows(10, 2, 2, C.byref(a), C.byref(cnt)) == -1


# ------------------------------------------------------------------------------------------------
# GPU (-m gpu): the same sharding logic with the real HIP engine
# ------------------------------------------------------------------------------------------------

def gpu_worker(rank, world, port, n, c, base_row, q):
    """two ranks share cuda:0 (the box has one GPU); the exchange runs over gloo, staged through host memory"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from shared_simd_scan_amd import ScanEngine
        from shared_simd_scan_amd.sharded import ShardedColumn, TorchExchange

        O = oracle()
        eng = ScanEngine(0)
        sc = ShardedColumn(n, c, engine=eng, base_row=base_row)
        assert isinstance(sc.exchange, TorchExchange)
        sc.generate("splitmix", 42)
        key = int(O.gen_values("splitmix", 1, c, 42, first=12345)[0])
        full, hits = sc.scan(key, dst=0)
        lo, hi = (1 << c) // 4, (1 << c) // 2
        full_r, hits_r = sc.scan_range(lo, hi, dst=0)
        full_p, hits_p = sc.scan_pipelined(key, dst=0, chunks=3)
        ids, nids = sc.select("between", lo, dst=0, b=hi)  # fused selection per shard, global ids gathered on the root
        local_bm, _ = eng.scan_range(lo, hi, sc.col) if sc.rows else (None, None)
        agg = sc.aggregate(mask=local_bm)
        vals_all = O.gen_values("splitmix", n, c, 42, first=base_row).astype(np.uint64)
        sel_all = vals_all[(vals_all >= lo) & (vals_all <= hi)]
        assert agg == ((int(sel_all.sum()), int(sel_all.shape[0]), int(sel_all.min()), int(sel_all.max())) if sel_all.shape[0] else (0, 0, None, 0)), agg
        if rank == 0:
            vals = O.gen_values("splitmix", n, c, 42, first=base_row)
            want_ids = np.nonzero((vals >= lo) & (vals <= hi))[0].astype(np.int64) + base_row
            assert nids == want_ids.shape[0] and ids.is_cuda and np.array_equal(ids.cpu().numpy(), want_ids)
            packed = O.pack(vals, c)
            ref, ref_hits = O.scan_eq(packed, n, c, key)
            ref_r, ref_hits_r = O.scan_range(packed, n, c, lo, hi)
            ok = (full.is_cuda and np.array_equal(full.cpu().numpy(), ref) and int(hits.item()) == ref_hits
                  and np.array_equal(full_p.cpu().numpy(), ref) and int(hits_p.item()) == ref_hits
                  and np.array_equal(full_r.cpu().numpy(), ref_r) and int(hits_r.item()) == ref_hits_r)
            q.put(("ok" if ok else "mismatch", sc.ranges))
        else:
            assert full is None and full_p is None and ids is None and nids == int(hits_r.item())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,base_row", [(8192 * 7 + 77, 9, 0), (2_000_003, 12, 7_000_000_000), (5, 12, 0)])
def test_sharded_scan_world2_real_engine(n, c, base_row):
    """ShardedColumn.scan / scan_range / scan_pipelined with ScanEngine under gloo, two ranks on one GPU; the second
    case is BASELINE config 5's shard shape (c = 12, splitmix, rows starting at 7e9)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=gpu_worker, args=(r, world, port, n, c, base_row, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    status, ranges = q.get(timeout=10)
    assert status == "ok", ranges


@pytest.mark.gpu
def test_c_abi_exchange_single_rank():
    """The RCCL entry points of the C ABI with a communicator of one rank (the box has one GPU and RCCL refuses two ranks
    on one device): bootstrap, gather to self at explicit offsets, all-reduce, sharded scan = local scan."""
    import ctypes as C

    from oracle import oracle
    from shared_simd_scan_amd import ScanEngine, lib

    O = oracle()
    L = lib()
    eng = ScanEngine(0)
    ident = (C.c_uint8 * 128)()
    assert L.mi355_comm_get_unique_id(ident) == 0, L.mi355_last_error()
    comm = C.c_void_p()
    assert L.mi355_comm_create(eng._ctx, 1, 0, ident, C.byref(comm)) == 0, L.mi355_last_error()
    w, r = C.c_int(), C.c_int()
    assert L.mi355_comm_info(comm, C.byref(w), C.byref(r)) == 0 and (w.value, r.value) == (1, 0)
    n, c = 8192 * 5 + 123, 12
    col = eng.generate("splitmix", n, c, 42)
    key = int(O.gen_values("splitmix", 1, c, 42, first=777)[0])
    nb = (n + 7) // 8
    local = torch.zeros(nb + 16, dtype=torch.uint8, device="cuda")
    full = torch.full((nb + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    hits = torch.zeros(1, dtype=torch.int64, device="cuda")
    rows = (C.c_uint64 * 1)(n)
    assert L.mi355_sharded_scan_eq_dev(eng._ctx, comm, col.data.data_ptr(), c, key, local.data_ptr(), rows, 0,
                                       full.data_ptr(), hits.data_ptr()) == 0, L.mi355_last_error()
    ref, ref_hits = O.scan_eq(col.data.cpu().numpy(), n, c, key)
    assert np.array_equal(full[:nb].cpu().numpy(), ref) and int(hits.item()) == ref_hits
    assert bool((full[nb:] == 0xEE).all().item())  # nothing written past the gathered bytes
    # explicit offsets
    sizes, offs = (C.c_uint64 * 1)(nb), (C.c_uint64 * 1)(32)
    full.fill_(0xEE)
    assert L.mi355_gather_bitmaps_at_dev(eng._ctx, comm, local.data_ptr(), sizes, offs, 0, full.data_ptr()) == 0
    assert np.array_equal(full[32: 32 + nb].cpu().numpy(), ref) and bool((full[:32] == 0xEE).all().item())
    # argument checks
    assert L.mi355_gather_bitmaps_dev(eng._ctx, comm, local.data_ptr(), sizes, 3, full.data_ptr()) == -1
    assert L.mi355_allreduce_hits_dev(eng._ctx, comm, hits.data_ptr(), 1) == 0
    assert int(hits.item()) == ref_hits
    assert L.mi355_comm_destroy(comm) == 0


def test_missing_librccl_is_an_error_not_a_crash():
    """a host without librccl: mi355_comm_get_unique_id / mi355_comm_create return MI355_E_COMM with dlopen's message
    (round 2 called dlerror() twice and handed std::string a null pointer).  MI355_RCCL_LIB forces the miss; a fresh
    process, because the library is looked up once per process."""
    import subprocess
    import sys

    code = (
        "import ctypes as C, sys\n"
        "from shared_simd_scan_amd import lib\n"
        "L = lib(); buf = (C.c_uint8 * 128)()\n"
        "rc = L.mi355_comm_get_unique_id(buf); msg = L.mi355_last_error().decode()\n"
        "print(rc, msg)\n"
        "sys.exit(0 if (rc != 0 and 'librccl not found' in msg and 'no-such-librccl' in msg) else 1)\n")
    env = dict(os.environ, MI355_RCCL_LIB="/nonexistent/no-such-librccl.so")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr


def _bench_2ranks(extra_env, extra_args=()):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_BACKEND="gloo", **extra_env)
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "40000000", "--steps", "5",
                          "--warmup", "2", *extra_args], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


@pytest.mark.gpu
def test_bench_two_ranks_prints_strong_headline_and_weak_beside_it():
    """`python bench.py --gpus 2` (gloo rehearsal: two ranks share the one GPU): `value` is ONE column split over the
    ranks, the weak figure sits beside it, per-rank kernel times and the exchange's own rank count are there"""
    res, line = _bench_2ranks({})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["rows_total"] == 40_000_000 and sum(line["config"]["rows_per_gpu"]) == 40_000_000
    assert all(r % 8192 == 0 for r in line["config"]["rows_per_gpu"][:-1])
    assert line["value"] == line["strong_values_per_s"] > 0 and line["weak_values_per_s"] > 0
    assert len(line["per_rank_kernel_ms"]) == 2 and all(x > 0 for x in line["per_rank_kernel_ms"])
    assert line["ranks_seen"] == 2 and line["comm_world"] == 2
    assert line["gather_ms"] > 0 and "gather_error" not in line
    assert line["hits"] == (line["config"]["rows_per_gpu"][0] + 1) // 5  # rank 0's rows i with i % 5 == 3


@pytest.mark.gpu
@pytest.mark.parametrize("mode,status", [("raise", 4), ("hang", 3), ("hang1", 3)])
def test_bench_failed_exchange_prints_the_line_and_exits_non_zero(mode, status):
    """a failed (raise) or stuck (hang on every rank / on rank 1 only) exchange step: the scan line is still printed and
    the run's exit status is NOT zero"""
    res, line = _bench_2ranks({"BENCH_FORCE_EXCHANGE_FAILURE": mode}, ("--gather-timeout", "15"))
    assert res.returncode != 0, res.stdout[-2000:]
    assert line is not None and line["value"] > 0 and line["gather_error"], res.stdout[-2000:] + res.stderr[-2000:]
    assert f"exit status {status}" in line["gather_error"] or status == 4

"""Row-range sharding of a packed column over the GPUs of one node, one process per GPU.

The column shards trivially: rows are independent, so rank r owns the contiguous rows
[r*rows_per_rank, (r+1)*rows_per_rank) and scans them with no communication.  Shard boundaries are multiples
of the scan tile (8192 rows), so every shard's packed slice starts 16-byte aligned on a whole value and its
bitmap slice on a whole byte.  The only exchange step is the final gather of the per-shard bitmaps to one rank
and the sum of the hit counts.

That exchange lives behind the C ABI (include/mi355_scan.h: mi355_comm_create, mi355_gather_bitmaps_dev,
mi355_allreduce_hits_dev -- direct RCCL calls over xGMI, grouped ncclSend / ncclRecv straight into the root's final
bitmap); `RcclExchange` below only calls those entry points.  torch.distributed is the host channel that hands the
RCCL unique id to every rank, and -- `TorchExchange` -- the transport of the CPU rehearsal of this module (gloo,
world_size 2, tests/test_sharded_gloo.py), where no GPU and no RCCL exist.

The reference has no multi-device code; this follows SURVEY 8e.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

SHARD_ALIGN = 8192  # rows; mi355_tile_values(c) for c <= 16, a multiple of it for wider columns


def shard_rows(n: int, world: int, align: int = SHARD_ALIGN) -> List[Tuple[int, int]]:
    """Split n rows into `world` contiguous [first, last) ranges whose boundaries are multiples of `align`
    (the last shard takes the ragged remainder; trailing shards may be empty for tiny n)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    per = -(-n // world)            # ceil
    per = -(-per // align) * align  # round up to the tile
    out = []
    for r in range(world):
        a = min(n, r * per)
        b = min(n, (r + 1) * per)
        out.append((a, b))
    return out


def bitmap_bytes(rows: int) -> int:
    return (rows + 7) // 8


# ---------------------------------------------------------------------------------------------------------------
# the exchange step
# ---------------------------------------------------------------------------------------------------------------

class ExchangeUnavailable(RuntimeError):
    """the RCCL exchange of the C ABI could not be set up -- raised on EVERY rank of the group together"""


class TorchExchange:
    """Exchange over a torch.distributed process group: gloo on CPU tensors (tests, rehearsals on a box with fewer
    GPUs than ranks -- device tensors are staged through host memory), or torch's own NCCL group.

    Point-to-point: every remote slice is received at its final offset of the root's buffer -- no padding of ragged
    shards to the largest one, no concatenation pass on the root."""

    name = "torch.distributed"

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.staged = dist.get_backend(group) == "gloo"

    def _global(self, r: int) -> int:
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def gather_at(self, local: torch.Tensor, sizes: Sequence[int], offsets: Sequence[int], dst: int, out: Optional[torch.Tensor],
                  engine=None) -> None:
        """rank r's sizes[r] bytes land at out[offsets[r]:] on `dst` (out: a tensor on the root, ignored elsewhere)"""
        if self.staged and local.is_cuda:
            local = local.cpu()
        mine = sizes[self.rank]
        if self.rank != dst:
            if mine:
                dist.send(local[:mine].contiguous(), dst=self._global(dst), group=self.group)
            return
        host = out
        if self.staged and out.is_cuda:  # rehearsal on a box with fewer GPUs than ranks: receive on the host, copy up
            host = torch.empty(out.numel(), dtype=torch.uint8)
        reqs = []
        for r in range(self.world):
            if r == dst:
                if mine:
                    host[offsets[r]: offsets[r] + mine] = local[:mine]
            elif sizes[r]:
                reqs.append(dist.irecv(host[offsets[r]: offsets[r] + sizes[r]], src=self._global(r), group=self.group))
        for q in reqs:
            q.wait()
        if host is not out:
            for r in range(self.world):
                if sizes[r]:
                    out[offsets[r]: offsets[r] + sizes[r]] = host[offsets[r]: offsets[r] + sizes[r]].to(out.device)

    def gather(self, local: torch.Tensor, sizes: Sequence[int], dst: int = 0, out: Optional[torch.Tensor] = None,
               engine=None) -> Optional[torch.Tensor]:
        total = sum(sizes)
        if self.rank == dst and (out is None or out.numel() < total):
            out = torch.empty(total, dtype=torch.uint8, device=local.device)
        offsets = [sum(sizes[:r]) for r in range(self.world)]
        self.gather_at(local, sizes, offsets, dst, out, engine=engine)
        return out[:total] if self.rank == dst else None

    def sum_hits(self, hits: torch.Tensor, engine=None) -> torch.Tensor:
        dev = hits.device
        total = hits.cpu().clone() if (self.staged and hits.is_cuda) else hits.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        return total.to(dev)

    def info(self) -> Tuple[int, int]:
        return self.world, self.rank


def _flag_all_ok(ok: bool, group=None) -> bool:
    """True iff `ok` on EVERY rank of the torch.distributed group (MIN all-reduce on the group's own device type)"""
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class RcclExchange:
    """The product path: the exchange entry points of the C ABI (direct RCCL over xGMI).  One communicator rank per
    process; the unique id travels over the torch.distributed group that launched the ranks.

    Construction is COLLECTIVE and every rank leaves it with the same verdict (ExchangeUnavailable on all or on none):
    a one-sided failure would otherwise leave peers blocked in a broadcast / in ncclCommInitRank, or mix transports.
      1. every rank checks that librccl loads and answers (mi355_comm_get_unique_id); the flags are MIN-reduced;
      2. rank 0 ALWAYS broadcasts -- its id (made in step 1), never an early exit in front of the broadcast;
      3. every rank creates its communicator rank; the outcomes are MIN-reduced again, and a communicator that exists on
         some ranks only is destroyed before ExchangeUnavailable is raised everywhere."""

    name = "mi355 C ABI (RCCL)"

    def __init__(self, engine, group=None):
        from . import _capi
        from ._capi import lib

        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._comm = None
        buf = (C.c_uint8 * _capi.COMM_ID_BYTES)()
        rc = lib().mi355_comm_get_unique_id(buf)
        why = None if rc == 0 else f"rank {self.rank}: {lib().mi355_last_error().decode()}"
        if not _flag_all_ok(rc == 0, group):
            raise ExchangeUnavailable(why or "librccl unavailable on another rank")
        ident = [bytes(buf) if self.rank == 0 else None]
        src = 0 if group is None else dist.get_global_rank(group, 0)
        dist.broadcast_object_list(ident, src=src, group=group)
        comm = C.c_void_p()
        idbuf = (C.c_uint8 * _capi.COMM_ID_BYTES).from_buffer_copy(ident[0])
        rc = lib().mi355_comm_create(engine._ctx, self.world, self.rank, idbuf, C.byref(comm))
        why = None if rc == 0 else f"rank {self.rank}: {lib().mi355_last_error().decode()}"
        if rc == 0:
            self._comm = comm
        if not _flag_all_ok(rc == 0, group):
            self.close()
            raise ExchangeUnavailable(why or "mi355_comm_create failed on another rank")

    def info(self) -> Tuple[int, int]:
        """(world, rank) as the communicator itself reports them (mi355_comm_info)"""
        from ._capi import check, lib

        w, r = C.c_int(), C.c_int()
        check(lib().mi355_comm_info(self._comm, C.byref(w), C.byref(r)))
        return w.value, r.value

    def close(self) -> None:
        from ._capi import lib

        if getattr(self, "_comm", None):
            lib().mi355_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gather_at(self, local: torch.Tensor, sizes: Sequence[int], offsets: Sequence[int], dst: int, out: Optional[torch.Tensor],
                  engine=None) -> None:
        from ._capi import check, lib

        arr = (C.c_uint64 * self.world)(*[int(x) for x in sizes])
        offs = (C.c_uint64 * self.world)(*[int(x) for x in offsets])
        check(lib().mi355_gather_bitmaps_at_dev(engine._ctx, self._comm, local.data_ptr() if local.numel() else None, arr, offs,
                                                dst, out.data_ptr() if self.rank == dst else None))

    def gather(self, local: torch.Tensor, sizes: Sequence[int], dst: int = 0, out: Optional[torch.Tensor] = None,
               engine=None) -> Optional[torch.Tensor]:
        from ._capi import check, lib

        total = sum(sizes)
        if self.rank == dst and (out is None or out.numel() < total):
            out = torch.empty(total, dtype=torch.uint8, device=local.device)
        arr = (C.c_uint64 * self.world)(*[int(x) for x in sizes])
        check(lib().mi355_gather_bitmaps_dev(engine._ctx, self._comm, local.data_ptr() if local.numel() else None, arr, dst,
                                             out.data_ptr() if self.rank == dst else None))
        return out[:total] if self.rank == dst else None

    def sum_hits(self, hits: torch.Tensor, engine=None) -> torch.Tensor:
        from ._capi import check, lib

        total = hits.clone()
        check(lib().mi355_allreduce_hits_dev(engine._ctx, self._comm, total.data_ptr(), total.numel()))
        return total


def make_exchange(engine=None, group=None):
    """RCCL through the C ABI when the ranks own distinct GPUs (process group backend "nccl"); the torch.distributed
    transport for gloo groups (CPU tests; several ranks sharing one GPU, which RCCL refuses).  MI355_EXCHANGE=torch|rccl
    overrides.  Collective: every rank returns the same kind of exchange, or every rank raises ExchangeUnavailable."""
    want = os.environ.get("MI355_EXCHANGE", "")
    backend = dist.get_backend(group)
    if want == "rccl" or (want != "torch" and backend == "nccl" and engine is not None and hasattr(engine, "_ctx")):
        return RcclExchange(engine, group)
    return TorchExchange(group)


def gather_bitmaps(local: torch.Tensor, dst: int = 0, out: Optional[torch.Tensor] = None,
                   sizes: Optional[List[int]] = None, group=None, exchange=None, engine=None) -> Optional[torch.Tensor]:
    """Gather per-shard bitmaps (uint8, shard r = bytes of rows of rank r) to rank `dst`: the concatenated bitmap on
    `dst`, None elsewhere.  `sizes` = bytes per rank (default: every rank sends local.numel() bytes)."""
    ex = exchange if exchange is not None else TorchExchange(group)
    if sizes is None:
        sizes = [local.numel()] * ex.world
    return ex.gather(local, sizes, dst=dst, out=out, engine=engine)


def sum_hits(local_hits: torch.Tensor, group=None, exchange=None, engine=None) -> torch.Tensor:
    """All ranks get the column-wide hit count(s)."""
    ex = exchange if exchange is not None else TorchExchange(group)
    return ex.sum_hits(local_hits, engine=engine)


# ---------------------------------------------------------------------------------------------------------------
# a column partitioned by row range
# ---------------------------------------------------------------------------------------------------------------

class ShardedColumn:
    """A packed column partitioned by row range over the ranks of a process group.

    `engine` is the per-shard scan: the HIP engine (ScanEngine; no CPU fallback).  `base_row` is the global row index
    of the column's row 0 (a column that is itself a slice of a larger one: generators hash the global index)."""

    def __init__(self, n: int, c: int, engine=None, group=None, base_row: int = 0, exchange=None):
        self.n, self.c, self.group, self.base_row = n, c, group, base_row
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ranges = shard_rows(n, self.world)
        self.first, self.last = self.ranges[self.rank]
        self.rows = self.last - self.first
        self.engine = engine
        self.exchange = exchange if exchange is not None else make_exchange(engine, group)
        self.col = None

    def generate(self, kind: str, param: int = 0) -> None:
        """every rank synthesises its own slice from the global row index (no upload, no scatter)"""
        self.col = self.engine.generate(kind, self.rows, self.c, param, first_row=self.base_row + self.first)

    def scan(self, key: int, dst: int = 0):
        bitmap, hits = self.engine.scan(key, self.col)
        return self._finish(bitmap, hits, dst)

    def scan_range(self, lo: int, hi: int, dst: int = 0):
        bitmap, hits = self.engine.scan_range(lo, hi, self.col)
        return self._finish(bitmap, hits, dst)

    def select(self, op: str, a: int, dst: int = 0, b: int = 0):
        """predicate -> the column's ascending GLOBAL row ids on rank `dst` (None elsewhere) and the total count (an int, on
        every rank).  Each rank runs the fused selection on its shard (no bitmap anywhere; ids are base_row + row index);
        the per-rank counts travel as ONE all-reduce of a one-hot vector; every rank's ids are then received at their
        final offset of the root's id list (shards are row ranges in rank order, so the concatenation is ascending)."""
        eng, ex, dev = self.engine, self.exchange, self._device()
        # ids cost 8 B each: start with room for 1 row in 16 (never the 8 B x rows of the worst case up front) and run the
        # shard again with the exact count when the predicate turns out denser (the count is the total whatever the capacity)
        cap = max(1, min(self.rows, max(1 << 16, self.rows // 16)))
        ids, cnt = eng.scan_select(op, a, self.col, capacity=cap, b=b, first_row=self.base_row + self.first)
        mine = int(cnt.reshape(-1)[0].item())
        if mine > cap:
            ids, cnt = eng.scan_select(op, a, self.col, capacity=mine, b=b, first_row=self.base_row + self.first)
            mine = int(cnt.reshape(-1)[0].item())
        # a count of UINT64_MAX (-1 as int64) is the engine's "selection failed" flag: it must not become a buffer size.
        # Decided collectively -- one more slot of the same all-reduce -- so that every rank raises, none is left in a send
        onehot = torch.zeros(self.world + 1, dtype=torch.int64, device=dev)
        onehot[self.rank] = max(mine, 0)
        onehot[self.world] = 1 if mine < 0 else 0
        summed = [int(x) for x in ex.sum_hits(onehot, engine=eng).cpu().tolist()]  # (the host needs them to size the receives)
        if summed[self.world]:
            raise RuntimeError(f"scan_select failed on {summed[self.world]} rank(s) (count = UINT64_MAX)")
        counts = summed[: self.world]
        sizes = [8 * k for k in counts]
        offsets = [sum(sizes[:r]) for r in range(self.world)]
        out = torch.empty(sum(counts), dtype=torch.int64, device=dev) if self.rank == dst else None
        local = ids.contiguous().view(torch.uint8)[: sizes[self.rank]]
        ex.gather_at(local, sizes, offsets, dst, out.view(torch.uint8) if out is not None else None, engine=eng)
        return out, sum(counts)

    def aggregate(self, mask: Optional[torch.Tensor] = None):
        """(sum, count, min, max) of the whole column over the rows of `mask` (this rank's slice of a result bitmap; None =
        every row), as Python ints on every rank: each rank's one-pass aggregate, then ONE all-reduce -- sums add, and the
        per-rank minima / maxima travel in a one-hot vector (min is None when no row counts)."""
        eng, ex, dev = self.engine, self.exchange, self._device()
        if self.rows:
            local = eng.aggregate(self.col, mask=mask).to(dev)
        else:
            local = torch.tensor([0, 0, -1, 0], dtype=torch.int64, device=dev)
        vec = torch.zeros(2 + 2 * self.world, dtype=torch.int64, device=dev)
        vec[0:2] = local[0:2]
        # (min = UINT64_MAX = -1 when the rank counted nothing: sent as -1, skipped below)
        vec[2 + 2 * self.rank] = local[2]
        vec[3 + 2 * self.rank] = local[3]
        tot = ex.sum_hits(vec, engine=eng).cpu().tolist()
        mins = [tot[2 + 2 * r] for r in range(self.world) if tot[2 + 2 * r] >= 0]
        maxs = [tot[3 + 2 * r] for r in range(self.world)]
        return int(tot[0]), int(tot[1]), (min(mins) if mins and tot[1] else None), (max(maxs) if tot[1] else 0)

    def scan_pipelined(self, key: int, dst: int = 0, chunks: int = 4):
        """Same result as scan().  The shard is scanned in `chunks` row ranges (boundaries at multiples of SHARD_ALIGN)
        and the gather of range i is enqueued as soon as its scan is: with the RCCL exchange on a side stream the
        transfer of range i over xGMI runs while range i+1 is scanned (the gather, not the scan, dominates a
        multi-GPU query).  Every piece is received at its final offset of the root's bitmap."""
        world, rank, eng, ex = self.world, self.rank, self.engine, self.exchange
        per_rank = max(b - a for a, b in self.ranges)
        per = -(-max(per_rank, 1) // max(1, chunks))
        per = -(-per // SHARD_ALIGN) * SHARD_ALIGN         # rows per piece, the same on every rank
        npieces = max(1, -(-per_rank // per))
        cb = per // 8                                       # bytes per full piece
        sizes = [bitmap_bytes(b - a) for a, b in self.ranges]
        offs = [sum(sizes[:r]) for r in range(world)]
        dev = self._device()
        final = torch.empty(sum(sizes), dtype=torch.uint8, device=dev) if rank == dst else None
        # the exchange runs on a side stream (its own context on the same device), ordered behind each piece's scan
        comm_eng, main, side = eng, None, None
        if isinstance(ex, RcclExchange):
            comm_eng, main, side = self._side_engine()
        hits_total = None
        keep = []
        for i in range(npieces):
            a, b = min(self.rows, i * per), min(self.rows, (i + 1) * per)
            piece = torch.empty(0, dtype=torch.uint8, device=dev)
            if b > a:
                bm, h = eng.scan(key, eng.slice_rows(self.col, a, b))
                hits_total = h.clone() if hits_total is None else hits_total + h
                piece = bm[: bitmap_bytes(b - a)]
            keep.append(piece)
            psizes = [bitmap_bytes(max(0, min(per, (self.ranges[r][1] - self.ranges[r][0]) - i * per))) for r in range(world)]
            poffs = [offs[r] + i * cb for r in range(world)]
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
            ex.gather_at(piece, psizes, poffs, dst, final, engine=comm_eng)
        if side is not None:
            main.wait_stream(side)
        if hits_total is None:
            hits_total = torch.zeros(1, dtype=torch.int64, device=dev)
        return final, ex.sum_hits(hits_total, engine=eng)

    def _side_engine(self):
        """(engine bound to a side stream, main torch stream, side torch stream) for overlapping the exchange with scans"""
        if getattr(self, "_side", None) is None:
            from .engine import ScanEngine

            side = torch.cuda.Stream(device=self.engine._dev)
            self._side = (ScanEngine(self.engine.device, stream=side), self.engine.stream, side)
        return self._side

    def _device(self):
        data = getattr(self.col, "data", None)
        if isinstance(data, torch.Tensor):
            return data.device
        return getattr(self.engine, "_dev", torch.device("cpu"))

    def _finish(self, bitmap, hits, dst):
        sizes = [bitmap_bytes(b - a) for a, b in self.ranges]
        full = self.exchange.gather(bitmap[: sizes[self.rank]], sizes, dst=dst, engine=self.engine)
        return full, self.exchange.sum_hits(hits, engine=self.engine)

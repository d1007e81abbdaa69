"""Row-range sharding of a packed column over the GPUs of one node, one process per GPU.

The column shards trivially: rows are independent, so rank r owns the contiguous rows
[r*rows_per_rank, (r+1)*rows_per_rank) and scans them with no communication.  Shard boundaries are multiples
of the scan tile (8192 rows), so every shard's packed slice starts 16-byte aligned on a whole value and its
bitmap slice on a whole byte.  The only exchange step is the final gather of the per-shard bitmaps to one rank
(RCCL over xGMI when the process group is "nccl"; the same code runs on "gloo" for CPU tests) and the sum of
the hit counts.

The reference has no multi-device code; this follows SURVEY 8e.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

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


def gather_bitmaps(local: torch.Tensor, dst: int = 0, out: Optional[torch.Tensor] = None,
                   sizes: Optional[List[int]] = None, group=None) -> Optional[torch.Tensor]:
    """Gather per-shard bitmaps (uint8, shard r = bytes of rows of rank r) to rank `dst`.

    Equal-sized shards use one dist.gather (RCCL: grouped send/recv into the root, all inbound xGMI links in
    parallel); ragged shards (`sizes` = bytes per rank) are padded to the largest and trimmed on the root.
    Returns the concatenated bitmap on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the N>1 path on a box with fewer GPUs than ranks: stage through host memory
        res = gather_bitmaps(local.cpu(), dst=dst, out=None, sizes=sizes, group=group)
        return res.to(local.device) if res is not None else None
    nbytes = local.numel()
    if sizes is None:
        sizes = [nbytes] * world
    biggest = max(sizes)
    if nbytes != biggest:
        padded = torch.zeros(biggest, dtype=torch.uint8, device=local.device)
        padded[:nbytes] = local
        local = padded
    if rank == dst:
        if out is None or out.numel() != biggest * world:
            out = torch.empty(biggest * world, dtype=torch.uint8, device=local.device)
        chunks = list(out.view(world, biggest).unbind(0))
        dist.gather(local, gather_list=chunks, dst=dst, group=group)
        if all(s == biggest for s in sizes):
            return out
        return torch.cat([chunks[r][: sizes[r]] for r in range(world)])
    dist.gather(local, gather_list=None, dst=dst, group=group)
    return None


def sum_hits(local_hits: torch.Tensor, group=None) -> torch.Tensor:
    """All ranks get the column-wide hit count(s)."""
    if local_hits.is_cuda and dist.get_backend(group) == "gloo":
        return sum_hits(local_hits.cpu(), group).to(local_hits.device)
    total = local_hits.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return total


class ShardedColumn:
    """A packed column partitioned by row range over the ranks of a process group.

    `local_scan(key, first_row, rows) -> (bitmap uint8[ceil(rows/8)], hits int64[1])` is the per-shard scan;
    by default it is the HIP engine (no CPU fallback)."""

    def __init__(self, n: int, c: int, engine=None, group=None):
        self.n, self.c, self.group = n, c, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ranges = shard_rows(n, self.world)
        self.first, self.last = self.ranges[self.rank]
        self.rows = self.last - self.first
        self.engine = engine
        self.col = None

    def generate(self, kind: str, param: int = 0) -> None:
        """every rank synthesises its own slice from the global row index (no upload, no scatter)"""
        self.col = self.engine.generate(kind, self.rows, self.c, param, first_row=self.first)

    def scan(self, key: int, dst: int = 0):
        bitmap, hits = self.engine.scan(key, self.col)
        return self._finish(bitmap, hits, dst)

    def scan_range(self, lo: int, hi: int, dst: int = 0):
        bitmap, hits = self.engine.scan_range(lo, hi, self.col)
        return self._finish(bitmap, hits, dst)

    def scan_pipelined(self, key: int, dst: int = 0, chunks: int = 4):
        """Same result as scan().  The shard is scanned in `chunks` row ranges (boundaries at multiples of SHARD_ALIGN)
        and the gather of range i is started asynchronously as soon as its scan is enqueued: with RCCL the transfer of
        range i over xGMI runs while range i+1 is scanned (the gather, not the scan, dominates a multi-GPU
        query).  Every rank sends `chunks` equal-sized pieces (short or empty ones are zero-padded); on the root, full
        pieces land directly in the final bitmap."""
        world, rank, eng = self.world, self.rank, self.engine
        per_rank = max(b - a for a, b in self.ranges)
        per = -(-max(per_rank, 1) // max(1, chunks))
        per = -(-per // SHARD_ALIGN) * SHARD_ALIGN         # rows per piece, the same on every rank
        npieces = max(1, -(-per_rank // per))
        cb = per // 8                                       # bytes per piece
        sizes = [bitmap_bytes(b - a) for a, b in self.ranges]
        offs = [sum(sizes[:r]) for r in range(world)]
        staged_via_host = None
        final, temps, works, keep = None, [], [], []
        hits_total = None
        for i in range(npieces):
            a, b = min(self.rows, i * per), min(self.rows, (i + 1) * per)
            piece = None
            if b > a:
                bm, h = eng.scan(key, eng.slice_rows(self.col, a, b))
                hits_total = h.clone() if hits_total is None else hits_total + h
                piece = bm[: bitmap_bytes(b - a)]
            if piece is None or piece.numel() != cb:
                padded = torch.zeros(cb, dtype=torch.uint8, device=piece.device if piece is not None else self._device())
                if piece is not None:
                    padded[: piece.numel()] = piece
                piece = padded
            if staged_via_host is None:
                staged_via_host = piece.is_cuda and dist.get_backend(self.group) == "gloo"
            if staged_via_host:
                piece = piece.cpu()  # rehearsal on a box with fewer GPUs than ranks: no overlap, same data path
            gather_list = None
            if rank == dst:
                if final is None:
                    final = torch.empty(sum(sizes), dtype=torch.uint8, device=piece.device)
                gather_list = []
                for r in range(world):
                    rows_r = self.ranges[r][1] - self.ranges[r][0]
                    valid = bitmap_bytes(max(0, min(per, rows_r - i * per)))
                    if valid == cb:
                        gather_list.append(final[offs[r] + i * cb: offs[r] + (i + 1) * cb])
                    else:
                        t = torch.empty(cb, dtype=torch.uint8, device=piece.device)
                        temps.append((t, offs[r] + i * cb, valid))
                        gather_list.append(t)
            keep.append(piece)
            works.append(dist.gather(piece, gather_list=gather_list, dst=dst, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if hits_total is None:
            hits_total = torch.zeros(1, dtype=torch.int64, device=self._device())
        if rank == dst:
            for t, off, valid in temps:
                if valid:
                    final[off: off + valid] = t[:valid]
            if staged_via_host:
                final = final.to(self._device())
        return (final if rank == dst else None), sum_hits(hits_total, self.group)

    def _device(self):
        data = getattr(self.col, "data", None)
        return data.device if isinstance(data, torch.Tensor) else torch.device("cpu")

    def _finish(self, bitmap, hits, dst):
        sizes = [bitmap_bytes(b - a) for a, b in self.ranges]
        full = gather_bitmaps(bitmap[: sizes[self.rank]], dst=dst, sizes=sizes, group=self.group)
        return full, sum_hits(hits, self.group)

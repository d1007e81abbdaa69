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

    def _finish(self, bitmap, hits, dst):
        sizes = [bitmap_bytes(b - a) for a, b in self.ranges]
        full = gather_bitmaps(bitmap[: sizes[self.rank]], dst=dst, sizes=sizes, group=self.group)
        return full, sum_hits(hits, self.group)

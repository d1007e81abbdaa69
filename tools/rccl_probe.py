#!/usr/bin/env python3
"""Probe: can the C ABI's RCCL exchange run with two ranks on ONE device?  (RCCL normally refuses duplicate GPUs.)
    python tools/rccl_probe.py        -> prints what happened; exit code 0 either way"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from shared_simd_scan_amd import ScanEngine
        from shared_simd_scan_amd.sharded import RcclExchange

        eng = ScanEngine(0)
        try:
            ex = RcclExchange(eng)
        except Exception as e:
            print(f"rank {rank}: RcclExchange failed: {e}", flush=True)
            return
        local = torch.full((1000 + rank,), rank + 1, dtype=torch.uint8, device="cuda")
        out = ex.gather(local, [1000, 1001], dst=0, engine=eng)
        h = ex.sum_hits(torch.tensor([rank + 5], dtype=torch.int64, device="cuda"), engine=eng)
        torch.cuda.synchronize()
        if rank == 0:
            print("gather ok:", bool((out[:1000] == 1).all() and (out[1000:] == 2).all()), "hits", int(h.item()), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        if p.is_alive():
            p.kill()
            print("probe: a rank hung (killed)")
    print("probe done, exit codes", [p.exitcode for p in ps])

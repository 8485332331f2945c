"""Sharding of reference views across the GPUs of a node.

Each reference view (cluster = 1 reference + N-1 source images) is an independent depth map; the
reference simply loops over clusters (mvsnet/inference.py:105-119).  Here one process drives one
GPU and takes clusters rank, rank+P, rank+2P, ... of the list sorted by (session, ref_index).
There is no data-path collective; the only communication is an optional gather of per-rank
counters at the end (gloo on CPU, RCCL on GPUs -- `backend="nccl"` is RCCL on ROCm).
"""
from __future__ import annotations

import os
from typing import List, Sequence


def rank_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin shard: items rank, rank+world, ... (balanced to within one item)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %d/%d" % (rank, world))
    return list(range(rank, n_items, world))


def shard(items: Sequence, rank: int, world: int) -> list:
    return [items[i] for i in shard_indices(len(items), rank, world)]


def init_process_group(backend=None):
    """Initialises torch.distributed when WORLD_SIZE > 1 (rendezvous on 127.0.0.1 by default).
    Returns the module or None for single-process runs."""
    rank, local_rank, world = rank_world()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def gather_counts(dist, value: float, device="cpu") -> List[float]:
    """All-gather one scalar per rank (e.g. depth maps written, seconds spent)."""
    if dist is None:
        return [float(value)]
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]

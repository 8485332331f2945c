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


def bind_device(local_rank=None):
    """Makes GPU `local_rank` (default: LOCAL_RANK) this process's current HIP device and returns its
    torch.device.  Must run before the first library call: kernels launch on the CURRENT device and
    the library creates its side streams / events there on first use (`_lib.ptr` refuses tensors of
    any other device).  Raises when the node has fewer GPUs than ranks instead of silently sharing one."""
    import torch
    if local_rank is None:
        local_rank = rank_world()[1]
    n = torch.cuda.device_count()
    local_rank = device_index_of(local_rank, n)
    if not (0 <= local_rank < n):
        raise RuntimeError("local rank %d needs cuda:%d but only %d GPU(s) are visible" % (local_rank, local_rank, n))
    torch.cuda.set_device(local_rank)
    return torch.device("cuda", local_rank)


def device_index_of(local_rank, n_visible):
    """The GPU index local rank `local_rank` drives.  MVS_GPUS=G (set by launch_ranks for --procs_per_gpu P > 1: G x P ranks)
    folds the ranks onto the FIRST G devices, ranks r, r+G, r+2G, ... sharing GPU r -- not onto however many devices the node
    happens to show (round 3 took `% device_count`, so `--gpus 1 --procs_per_gpu 3` on an 8-GPU node used GPUs 0, 1 and 2).
    MVS_ALLOW_SHARED_GPU additionally folds onto the visible devices (rehearsing N ranks on a box with fewer GPUs)."""
    g = int(os.environ.get("MVS_GPUS", "0") or 0)
    if g > 0:
        local_rank = local_rank % g
    if n_visible > 0 and os.environ.get("MVS_ALLOW_SHARED_GPU"):
        local_rank = local_rank % n_visible
    return local_rank


def launch_ranks(script_args, n_ranks, module=None, gpus=None):
    """Starts `n_ranks` one-GPU worker processes of `script_args` (or of `-m module`) under torch.distributed.run on this
    node and returns its exit code (non-zero when any rank failed).  For entry points started plainly as
    `python bench.py --gpus N` / `python -m mvsnet_amd.inference --gpus N`: the caller must not have touched the GPU
    (the parent stays GPU-less, every rank binds its own device), the ranks inherit stdout / stderr, rank 0 prints.
    `gpus` < n_ranks: several ranks per GPU on the first `gpus` devices (MVS_GPUS, see device_index_of)."""
    import socket
    import subprocess
    import sys
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:      # a free rendezvous port on the loopback address
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
           "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += (["-m", module] if module else []) + list(script_args)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    if gpus is not None and 0 < int(gpus) < int(n_ranks):
        env["MVS_GPUS"] = str(int(gpus))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // int(n_ranks))))
    return subprocess.call(cmd, env=env)


def init_process_group(backend=None):
    """Initialises torch.distributed when WORLD_SIZE > 1 (rendezvous on 127.0.0.1 by default).
    Returns the module or None for single-process runs.  With the RCCL backend the process is bound
    to its GPU first and the group is created with that `device_id`."""
    rank, local_rank, world = rank_world()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:        # MVS_DIST_BACKEND=gloo: rehearsals with several ranks per GPU (RCCL wants one GPU per rank)
        backend = os.environ.get("MVS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=bind_device(local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def gather_counts(dist, value: float, device="cpu") -> List[float]:
    """All-gather one scalar per rank (e.g. depth maps written, seconds spent)."""
    if dist is None:
        return [float(value)]
    import torch
    if dist.get_backend() != "nccl":
        device = "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]

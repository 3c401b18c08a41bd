"""Multi-GPU sharding of the recurrent path: one process per GPU, sequences
partitioned statically, no exchange while frames are computed.

The reference has no distributed code at all (its only parallelism is a
single-process ``torch.nn.DataParallel`` wrapper that is a no-op at
validation batch size, networks/__init__.py:110-113), so this is new, not a
translation: independent video sequences share nothing but the read-only
weights, the recurrence is strictly inside a sequence
(models/recurrent_model.py:233-345), hence the only collectives are the
barrier / max that bracket a timed region and ONE all-gather that collates the
per-frame metrics afterwards (RCCL over xGMI on the GPU box, gloo in the CPU
tests -- same code).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """-> (rank, local_rank, world, dist_module_or_None).  Reads the
    torch.distributed.run environment.  Whenever a launcher has set RANK and
    WORLD_SIZE the process group IS initialised -- at world size 1 too, so that
    `torch.distributed.run --nproc-per-node 1` runs the same RCCL init and the
    same collectives on device tensors as the 8-GPU job does; only a plain
    `python bench.py` (no launcher environment) runs without a group."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        if world != 1:
            raise RuntimeError("WORLD_SIZE is set without RANK: start the ranks with torch.distributed.run")
        return rank, local_rank, world, None
    # the host driver only supports dmabuf IPC; RCCL's intra-node transport needs this before its first call
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world, dist


def shard_sequences(n_total: int, rank: int, world: int) -> range:
    """Static block partition ``sequence s -> rank s // ceil(n/world)``."""
    per = -(-n_total // world)
    lo = min(rank * per, n_total)
    return range(lo, min(lo + per, n_total))


def barrier(dist, device: Optional[torch.device] = None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, dist, device: Optional[torch.device] = None) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_metrics(local: torch.Tensor, dist) -> torch.Tensor:
    """The one collate collective: [n_local, ...] per rank -> [world*n_local, ...]
    in rank (= sequence) order.  Every rank must pass the same shape."""
    if dist is None:
        return local
    parts: List[torch.Tensor] = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local.contiguous())
    return torch.cat(parts, 0)


def count_ranks(dist, device: Optional[torch.device] = None) -> int:
    """How many ranks take part, counted by the collective itself (a sum of ones)."""
    t = torch.ones(1, dtype=torch.int64, device=device if device is not None else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_outputs(local: torch.Tensor, dist) -> torch.Tensor:
    """Collate the output frames: [...] per rank -> [world, ...] on every rank, ONE all-gather
    (RCCL picks the direct one-hop algorithm over xGMI for a single large message per peer)."""
    if dist is None:
        return local[None]
    world = dist.get_world_size()
    local = local.contiguous()
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "gloo":
        dist.all_gather(list(out.unbind(0)), local)
    else:
        dist.all_gather_into_tensor(out, local)
    return out

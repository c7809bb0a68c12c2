"""Multi-GPU plumbing for the embed + Ward path: one process per GPU, torch.distributed for the
rendezvous and the one real exchange step of the path -- assembling E on every rank.

* embed shards naturally: rank r owns the contiguous image range shard_range(n_total, r, world); no collective.
* E is assembled with ONE all-gather (backend "nccl" == RCCL over xGMI on MI355X; "gloo" on CPU in the tests).
* Ward runs on rank 0 over the gathered E (BASELINE.json configs[2]: "tiled Ward distance on GPU0").

Nothing here computes on tensors: device work stays in libimageclust_hip.so.
"""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, rank: int, world: int, device=None):
    if world <= 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")  # the container hostname may not resolve
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous index range [lo, hi) of rank's images; ragged totals give the first (n_total % world) ranks one more."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_embeddings(E_local: torch.Tensor, n_total: int, rank: int, world: int) -> torch.Tensor:
    """All-gather row shards of E (shard_range order) into the full n_total x D matrix on every rank."""
    if world <= 1:
        return E_local
    d = E_local.shape[1]
    base, extra = divmod(n_total, world)
    if extra == 0:
        out = torch.empty((n_total, d), dtype=E_local.dtype, device=E_local.device)
        dist.all_gather_into_tensor(out, E_local.contiguous())
        return out
    # ragged shards: pad to the largest shard, gather, then drop the padding rows
    cap = base + 1
    pad = torch.zeros((cap, d), dtype=E_local.dtype, device=E_local.device)
    pad[: E_local.shape[0]] = E_local
    buf = torch.empty((world * cap, d), dtype=E_local.dtype, device=E_local.device)
    dist.all_gather_into_tensor(buf, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(buf[r * cap: r * cap + (hi - lo)])
    return torch.cat(parts, 0)


def max_over_ranks(seconds: float, device=None) -> float:
    """bench.py's timing rule: the job takes as long as its slowest rank."""
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_cluster_ids(cluster_id: torch.Tensor, member_rank: torch.Tensor, src: int = 0):
    """Rank 0 clusters; every rank may need the ids of its own shard."""
    if dist.is_initialized():
        dist.broadcast(cluster_id, src)
        dist.broadcast(member_rank, src)
    return cluster_id, member_rank


def barrier():
    if dist.is_initialized():
        dist.barrier()

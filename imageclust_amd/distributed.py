"""Multi-GPU plumbing for the embed + Ward path: one process per GPU, torch.distributed for the
rendezvous and the one real exchange step of the path -- assembling E on every rank.

* embed shards naturally: rank r owns the contiguous image range shard_range(n_total, r, world); no collective.
* E is assembled with ONE all-gather (backend "nccl" == RCCL over xGMI on MI355X; "gloo" on CPU in the tests).
* the initial distance matrix (clustering.go:61-73) is built on ALL ranks: rank r computes an area-balanced run of
  128-row tile rows (one contiguous span of the packed triangle) and sends it to rank 0, which receives it straight into its
  triangle ("tiles computed on 8 GPUs, scattered to GPU0 over xGMI"); rank 0 computes its own run
  meanwhile.  No data-path collective: 7 point-to-point sends, one per xGMI link into GPU 0.
* the exact merge loop runs on rank 0 (BASELINE.json configs[2]: "tiled Ward distance on GPU0").

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


class _DeviceSpan:
    """Lets torch address `count` floats of engine-owned device memory without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def tile_plan(n: int, world: int):
    """[(row_lo, row_hi, float_off, float_cnt)] per rank: who computes which rows of the initial distance matrix."""
    from . import _lib

    plan = []
    for r in range(world):
        lo, hi = _lib.ward_rows_partition(n, world, r)
        off, cnt = _lib.ward_span(lo, hi)
        plan.append((lo, hi, off, cnt))
    return plan


SPAN_CHUNK = 1 << 28  # floats per point-to-point message (1 GiB): a rank's span can exceed 2^31 elements (40 GB at world 2)


def exchange_spans(rank: int, world: int, plan, my_span, recv_buffer, chunk: int = 0):
    """The one exchange of the distance build: every rank r > 0 sends its span to rank 0 in messages of at most `chunk`
    floats (default SPAN_CHUNK).  On rank 0 recv_buffer(r) returns the tensor the span of rank r lands in (for the GPU path: a
    view of the triangle itself).  Returns the pending requests (rank 0) so the caller can compute its own rows while the
    transfers run; messages of one peer are matched in order, peers proceed concurrently (one xGMI link each)."""
    if world <= 1:
        return []
    chunk = int(chunk) if chunk else SPAN_CHUNK
    if rank != 0:
        cnt = plan[rank][3]
        for off in range(0, cnt, chunk):
            dist.send(my_span[off:min(off + chunk, cnt)], dst=0)
        return []
    reqs = []
    bufs = {r: recv_buffer(r) for r in range(1, world) if plan[r][3] > 0}
    nmsg = max((plan[r][3] + chunk - 1) // chunk for r in range(world))
    for m in range(nmsg):  # round-robin over the peers so that every link has a receive posted from the start
        for r, buf in bufs.items():
            off = m * chunk
            if off < plan[r][3]:
                reqs.append(dist.irecv(buf[off:min(off + chunk, plan[r][3])], src=r))
    return reqs


def cluster_with_distributed_tiles(ctx, E_full: torch.Tensor, min_size: int, max_size: int, rank: int, world: int, update=0, staged=False):
    """PerformClusteringWithConstraints over `world` GPUs: distance rows on every rank, merge loop on rank 0.
    Returns (cluster_id, member_rank, n_clusters) on rank 0 and None elsewhere.  staged=True moves the spans through host
    memory (gloo rehearsal on a box whose ranks share one GPU)."""
    n, d = E_full.shape
    plan = tile_plan(n, world)
    lo, hi, _, cnt = plan[rank]
    if rank != 0:
        span = torch.empty(max(cnt, 1), dtype=torch.float32, device=E_full.device)
        if cnt:
            ctx.ward_distance_rows_dev(E_full.data_ptr(), n, d, lo, hi, span.data_ptr())
        exchange_spans(rank, world, plan, span[:cnt].cpu() if staged else span[:cnt], None)
        return None
    ctx.ward_prepare(n, d)
    views = {}

    def recv_buffer(r):
        if staged:
            views[r] = torch.empty(plan[r][3], dtype=torch.float32)
        else:
            ptr, c = ctx.ward_span_ptr(plan[r][0], plan[r][1])
            views[r] = torch.as_tensor(_DeviceSpan(ptr, c), device=E_full.device)
        return views[r]

    reqs = exchange_spans(rank, world, plan, None, recv_buffer)
    if cnt:  # rank 0's own rows, written into its triangle while the other spans arrive
        ptr, _ = ctx.ward_span_ptr(lo, hi)
        ctx.ward_distance_rows_dev(E_full.data_ptr(), n, d, lo, hi, ptr)
    for q in reqs:
        q.wait()
    if staged:
        for r, t in views.items():
            dev = t.to(E_full.device)
            ctx.ward_deposit_dev(plan[r][0], plan[r][1], dev.data_ptr())
    torch.cuda.synchronize()
    return ctx.cluster_prefilled_dev(E_full.data_ptr(), n, d, min_size, max_size, 0, 0, update)


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

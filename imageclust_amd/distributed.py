"""Multi-GPU plumbing for the embed + Ward path: one process per GPU, torch.distributed for the
rendezvous and the one real exchange step of the path -- assembling E on every rank.

* embed shards naturally: rank r owns the contiguous image range shard_range(n_total, r, world); no collective.
* E is assembled with ONE all-gather (backend "nccl" == RCCL over xGMI on MI355X; "gloo" on CPU in the tests).
* the initial distance matrix (clustering.go:61-73) can be built on ALL ranks (cluster_with_distributed_tiles): rank r computes
  an area-balanced run of 128-row tile rows (one contiguous span of the packed triangle) and sends it to rank 0 in pieces of
  whole rows; rank 0 receives every piece into one of two bounded landing buffers per peer and lays it out into its distance
  matrix at once (icl_ward_unpack_spans_dev), so it never holds more than the 4 n^2-byte matrix plus the landing buffers
  ("tiles computed on 8 GPUs, scattered to GPU0 over xGMI"); its own run comes from the matrix-core bounds
  inside the cluster call.  No data-path collective: 7 point-to-point streams, one per xGMI link into GPU 0.
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


SPAN_CHUNK = 1 << 26  # floats per point-to-point message (256 MiB): the size of a landing buffer on rank 0


def row_pieces(row_lo: int, row_hi: int, max_floats: int):
    """Cuts the rows [row_lo, row_hi) into runs of WHOLE rows of at most max_floats floats each (at least one row per run):
    [(r0, r1, float offset inside the span, floats)].  Sender and receiver derive the same cut."""
    from . import _lib

    base = _lib.ward_span(0, row_lo)[1]
    out, r0 = [], row_lo
    while r0 < row_hi:
        # rows near r hold ~r floats: first guess by the row length, then adjust
        r1 = min(row_hi, r0 + max(1, max_floats // max(r0 + 4, 4)))
        while r1 < row_hi and _lib.ward_span(r0, r1 + 1)[1] <= max_floats:
            r1 += 1
        while r1 > r0 + 1 and _lib.ward_span(r0, r1)[1] > max_floats:
            r1 -= 1
        out.append((r0, r1, _lib.ward_span(0, r0)[1] - base, _lib.ward_span(r0, r1)[1]))
        r0 = r1
    return out


def send_pieces(span: torch.Tensor, row_lo: int, row_hi: int, chunk: int = 0):
    """Rank r > 0: its span (rows [row_lo, row_hi) of the packed triangle) to rank 0, one message per run of whole rows."""
    for _, _, off, c in row_pieces(row_lo, row_hi, int(chunk) if chunk else SPAN_CHUNK):
        dist.send(span[off:off + c], dst=0)


def receive_pieces(plan, world: int, make_buffer, deliver, chunk: int = 0):
    """Rank 0: receives every other rank's span piece by piece.  make_buffer(floats) allocates a landing buffer (two per peer, so
    piece k + 1 of a peer is in flight while piece k is consumed); deliver(row_lo, row_hi, tensor) consumes a landed piece and
    returns once the buffer may be overwritten.  Peers proceed concurrently (one xGMI link each), messages of one peer arrive in
    order.  Rank 0 never holds more than 2 x (world - 1) landing buffers of at most `chunk` floats (or one row)."""
    chunk = int(chunk) if chunk else SPAN_CHUNK
    pieces = {r: row_pieces(plan[r][0], plan[r][1], chunk) for r in range(1, world) if plan[r][3] > 0}
    land = {r: [make_buffer(max(c for _, _, _, c in p)) for _ in range(min(2, len(p)))] for r, p in pieces.items()}
    pend = {r: [] for r in pieces}  # per peer: [(piece index, request)]
    nxt = {r: 0 for r in pieces}

    def post(r):
        k = nxt[r]
        if k < len(pieces[r]):
            pend[r].append((k, dist.irecv(land[r][k % len(land[r])][:pieces[r][k][3]], src=r)))
            nxt[r] = k + 1

    for r in pieces:  # every link has receives posted from the start
        for _ in land[r]:
            post(r)
    while any(pend.values()):
        for r in pieces:  # round-robin over the peers
            if not pend[r]:
                continue
            k, req = pend[r].pop(0)
            req.wait()
            r0, r1, _, c = pieces[r][k]
            deliver(r0, r1, land[r][k % len(land[r])][:c])
            post(r)


def cluster_with_distributed_tiles(ctx, E_full: torch.Tensor, min_size: int, max_size: int, rank: int, world: int, update=0, staged=False,
                                   chunk: int = 0):
    """PerformClusteringWithConstraints over `world` GPUs: distance rows on every rank, merge loop on rank 0.
    Returns (cluster_id, member_rank, n_clusters) on rank 0 and None elsewhere.  staged=True moves the pieces through host
    memory (gloo rehearsal on a box whose ranks share one GPU).  chunk: floats per message (default SPAN_CHUNK)."""
    n, d = E_full.shape
    plan = tile_plan(n, world)
    lo, hi, _, cnt = plan[rank]
    if rank != 0:
        if cnt:
            span = torch.empty(cnt, dtype=torch.float32, device=E_full.device)
            ctx.ward_distance_rows_dev(E_full.data_ptr(), n, d, lo, hi, span.data_ptr())
            send_pieces(span.cpu() if staged else span, lo, hi, chunk)
        return None
    ctx.ward_prepare(n, d)

    def deliver(r0, r1, buf):
        dev = buf.to(E_full.device) if staged else buf
        torch.cuda.current_stream().synchronize()  # the transport's write into the landing buffer is complete
        ctx.ward_unpack_spans_dev([(r0, r1, dev.data_ptr())])  # returns when the rows are in place

    receive_pieces(plan, world, lambda c: torch.empty(c, dtype=torch.float32, device="cpu" if staged else E_full.device), deliver, chunk)
    return ctx.cluster_prefilled_dev(E_full.data_ptr(), n, d, min_size, max_size, lo, hi, update)


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

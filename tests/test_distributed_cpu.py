"""world_size-2 (and 3, ragged) gloo tests of the N>1 plumbing (imageclust_amd/distributed.py) on CPU: sharding,
the all-gather that assembles E in shard order, the max-over-ranks timing rule and the id broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from imageclust_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, d, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    D.init("gloo", rank, world)
    full = torch.arange(n_total * d, dtype=torch.float32).reshape(n_total, d)  # the matrix a 1-GPU run would produce
    lo, hi = D.shard_range(n_total, rank, world)
    E = D.gather_embeddings(full[lo:hi].clone(), n_total, rank, world)
    ok_gather = bool(torch.equal(E, full))
    tmax = D.max_over_ranks(1.0 + rank)
    cid = torch.arange(n_total, dtype=torch.int32) if rank == 0 else torch.zeros(n_total, dtype=torch.int32)
    mr = torch.ones(n_total, dtype=torch.int32) * (7 if rank == 0 else 0)
    D.broadcast_cluster_ids(cid, mr, 0)
    D.barrier()
    q.put((rank, ok_gather, tmax, bool(torch.equal(cid, torch.arange(n_total, dtype=torch.int32))), int(mr[0])))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 10), (2, 11), (3, 10)])
def test_gather_and_timing_gloo(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for _, ok_gather, tmax, ok_ids, mr0 in res:
        assert ok_gather and ok_ids and mr0 == 7
        assert tmax == float(world)  # slowest rank: 1.0 + (world-1)


def test_shard_ranges_partition():
    for n in [0, 1, 7, 10000, 100003]:
        for w in [1, 2, 3, 8]:
            r = [D.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


# ---- distance tiles over ranks (imageclust_amd/distributed.py: tile_plan / row_pieces / send_pieces / receive_pieces) ----
def _pack_rows(D, lo, hi):
    """Rows [lo, hi) of a dense matrix in the engine's packed-triangle layout: row r holds D[r, :r], padded to 4 floats."""
    out = []
    for r in range(lo, hi):
        row = np.zeros((r + 3) // 4 * 4, np.float32)
        row[:r] = D[r, :r]
        out.append(row)
    return np.concatenate(out) if out else np.zeros(0, np.float32)


def test_tile_plan_covers_the_triangle_with_balanced_area():
    from imageclust_amd import _lib

    for n in [0, 1, 127, 128, 129, 1000, 10000, 100000, 250000]:
        for w in [1, 2, 3, 8]:
            plan = D.tile_plan(n, w)
            assert plan[0][0] == 0 and plan[-1][1] == n
            off = 0
            for (lo, hi, o, c) in plan:
                assert (lo % 128 == 0 or lo == n) and (hi % 128 == 0 or hi == n) and lo <= hi
                assert o == off  # spans tile the packed triangle back to back
                off += c
            assert off == _lib.ward_span(0, n)[1] == sum((r + 3) // 4 * 4 for r in range(n))
            if n >= 100000:  # equal AREA, not equal row counts: the first rank gets ~sqrt(1/w) of the rows
                areas = [c for (_, _, _, c) in plan]
                assert max(areas) / (sum(areas) / w) < 1.05
                assert plan[0][1] - plan[0][0] > 2 * (plan[-1][1] - plan[-1][0]) or w == 1


def _span_worker(rank, world, port, n, q, chunk=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    D.init("gloo", rank, world)
    from oracle import oracle as O

    rng = np.random.default_rng(7)
    E = rng.standard_normal((n, 6)).astype(np.float32)  # every rank holds the gathered E
    full = O.initial_distance_matrix(E)
    plan = D.tile_plan(n, world)
    lo, hi, off, cnt = plan[rank]
    mine = torch.from_numpy(_pack_rows(full, lo, hi))  # stands in for icl_ward_distance_rows_dev on this rank's GPU
    assert mine.numel() == cnt
    ok = True
    if rank == 0:
        from imageclust_amd import _lib

        tri = torch.zeros(sum(p[3] for p in plan), dtype=torch.float32)
        tri[off:off + cnt] = mine
        landed = []  # landing buffers handed out: at most two per peer, each at most one piece long

        def make_buffer(c):
            landed.append(c)
            return torch.empty(c, dtype=torch.float32)

        def deliver(r0, r1, buf):  # stands in for icl_ward_unpack_spans_dev: whole rows, at their place in the triangle
            o, c = _lib.ward_span(0, r0)[1], _lib.ward_span(r0, r1)[1]
            assert buf.numel() == c
            tri[o:o + c] = buf

        D.receive_pieces(plan, world, make_buffer, deliver, chunk)
        ok = bool(torch.equal(tri, torch.from_numpy(_pack_rows(full, 0, n))))  # rank 0 has seen exactly the packed triangle
        ok = ok and len(landed) <= 2 * (world - 1) and (not chunk or max(landed) <= max(chunk, (n + 3) // 4 * 4))
    else:
        if cnt:
            D.send_pieces(mine, lo, hi, chunk)
    D.barrier()
    q.put((rank, ok))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,n,chunk", [(2, 300, 0), (3, 700, 0), (2, 129, 0), (3, 700, 4099), (2, 300, 1)])
def test_span_exchange_assembles_the_packed_triangle_gloo(world, n, chunk):  # chunk > 0: spans cut into several runs of whole rows
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_span_worker, args=(r, world, port, n, q, chunk)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)

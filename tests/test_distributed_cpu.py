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

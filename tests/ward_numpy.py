"""Independent numpy-float32 re-restatement of clustering.go used to CHECK THE ORACLE (it is not the oracle).

Unlike oracle/ward_ref.c (literal, position-compacted) this one uses the creation-id formulation the GPU
engine relies on: static ids, a static size mask instead of the MaxFloat32 ban,
lexicographic (value, larger id, smaller id) selection.  Agreement of the two on random, tie-heavy and
constrained inputs is the evidence that the reformulation is exactly equivalent to the reference scan.
"""
import math

import numpy as np

MAXF = np.float32(np.finfo(np.float32).max)


def ward(ca, sa, cb, sb):
    diff = (ca - cb).astype(np.float32)
    s = np.float32(0)
    sq = (diff * diff).astype(np.float32)
    for v in sq:  # sequential fp32 accumulation, clustering.go:152-155
        s = np.float32(s + v)
    num = np.float32(sa * sb)
    den = np.float32(sa + sb)
    return np.float32(np.float32(num / den) * s)


def calc_k(total, mn, mx):
    if mn < 1 or mx < 1 or total < mn:
        return None
    lo = int(math.ceil(total / mx))
    hi = int(math.floor(total / mn))
    if lo > hi:
        return None
    return lo if lo == hi else (lo + hi) // 2


def cluster_creation_id(E, mn, mx):
    E = np.asarray(E, np.float32)
    n, d = E.shape
    k = calc_k(n, mn, mx)
    if k is None:
        return None
    M = 2 * n
    cent = {i: E[i].copy() for i in range(n)}
    size = {i: 1 for i in range(n)}
    members = {i: [i] for i in range(n)}
    alive = set(range(n))
    dist = {}
    for i in range(n):
        for j in range(i):
            dist[(i, j)] = ward(cent[i], 1, cent[j], 1)
    nxt = n
    while len(alive) > k:
        best = None
        for (i, j), v in dist.items():
            if i not in alive or j not in alive:
                continue
            if size[i] + size[j] > mx:
                continue
            if not (v < MAXF):
                continue
            key = (v, i, j)
            if best is None or key < best:
                best = key
        if best is None:
            break
        _, i, j = best
        c = nxt
        nxt += 1
        fa, fb, fs = np.float32(size[i]), np.float32(size[j]), np.float32(size[i] + size[j])
        cent[c] = (((fa * cent[i]).astype(np.float32) + (fb * cent[j]).astype(np.float32)).astype(np.float32) / fs).astype(np.float32)
        size[c] = size[i] + size[j]
        members[c] = members[i] + members[j]
        alive.discard(i)
        alive.discard(j)
        for x in alive:
            dist[(c, x)] = ward(cent[x], size[x], cent[c], size[c])
        alive.add(c)
    cid = np.full(n, -1, np.int32)
    rank = np.full(n, -1, np.int32)
    nid = 0
    for c in sorted(alive):
        if size[c] < mn:
            continue
        for r, m in enumerate(members[c]):
            cid[m] = nid
            rank[m] = r
        nid += 1
    return cid, rank, nid

"""Host-side mirror of /root/reference/internal/clustering/clustering.go over the HIP engine.

Same exported names, argument meaning and error behaviour as the Go package, so the parity tests read like
tests of the reference.  Every arithmetic result comes from libimageclust_hip.so through its C-ABI
(include/imageclust.h); the functions that only re-arrange lists (NewCluster, RemoveClusters,
RemoveRowsAndColumns) are plain data-structure code exactly as in the Go file.  There is no CPU fallback:
without the built library (or without a gfx950 device) these functions raise.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib

_default_ctx: Optional[_lib.Context] = None


def default_context() -> _lib.Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = _lib.Context(0)
    return _default_ctx


def set_default_context(ctx: Optional[_lib.Context]):
    global _default_ctx
    _default_ctx = ctx


@dataclass
class Cluster:
    """clustering.go:11-15"""
    Indices: List[int] = field(default_factory=list)
    Size: int = 0
    Centroid: np.ndarray = None


def NewCluster(index: int, embedding) -> Cluster:
    """clustering.go:18-26: singleton with a COPY of the embedding."""
    return Cluster(Indices=[int(index)], Size=1, Centroid=np.array(embedding, dtype=np.float32, copy=True))


def MergeClusters(a: Cluster, b: Cluster, ctx: Optional[_lib.Context] = None) -> Cluster:
    """clustering.go:29-47: members a++b, centroid (float(sa)*Ca + float(sb)*Cb)/float(sa+sb) on the GPU."""
    ctx = ctx or default_context()
    return Cluster(Indices=list(a.Indices) + list(b.Indices), Size=a.Size + b.Size,
                   Centroid=ctx.merge_centroid(a.Centroid, a.Size, b.Centroid, b.Size))


def RemoveClusters(clusters: List[Cluster], i: int, j: int) -> List[Cluster]:
    """clustering.go:51-58: order-preserving removal of positions i and j."""
    if i > j:
        i, j = j, i
    return clusters[:i] + clusters[i + 1:j] + clusters[j + 1:]


def ComputeInitialDistanceMatrix(clusters: List[Cluster], ctx: Optional[_lib.Context] = None) -> np.ndarray:
    """clustering.go:61-73 -> n x n fp32, symmetric, zero diagonal (exact Ward tile kernel)."""
    ctx = ctx or default_context()
    n = len(clusters)
    if n == 0:
        return np.zeros((0, 0), np.float32)
    C = np.stack([np.asarray(c.Centroid, np.float32) for c in clusters])
    sizes = np.array([c.Size for c in clusters], np.int32)
    return ctx.ward_distance_matrix(C, sizes)


def RemoveRowsAndColumns(matrix: np.ndarray, i: int, j: int) -> np.ndarray:
    """clustering.go:100-116"""
    keep = [k for k in range(matrix.shape[0]) if k != i and k != j]
    return matrix[np.ix_(keep, keep)]


def WardDistance(a: Cluster, b: Cluster, ctx: Optional[_lib.Context] = None) -> np.float32:
    """clustering.go:136-145 for one pair (a 2x2 call of the distance tile)."""
    return ComputeInitialDistanceMatrix([a, b], ctx)[1, 0]


def UpdateDistanceMatrix(distanceMatrix: np.ndarray, clusters: List[Cluster], newCluster: Cluster, removedIdx1: int,
                         removedIdx2: int, ctx: Optional[_lib.Context] = None) -> np.ndarray:
    """clustering.go:76-96: drop two rows/columns, append the new cluster's row/column computed from centroids
    (icl_update_distance_matrix).  `clusters` is the list AFTER removal and append (newCluster last), as at
    clustering.go:244; newCluster is accepted for signature parity and must be clusters[-1]."""
    ctx = ctx or default_context()
    C = np.stack([np.asarray(c.Centroid, np.float32) for c in clusters])
    sizes = np.array([c.Size for c in clusters], np.int32)
    return ctx.update_distance_matrix(np.asarray(distanceMatrix, np.float32), C, sizes, removedIdx1, removedIdx2)


def DotFloat32(a, b) -> np.float32:
    """clustering.go:148-157: in-order fp32 dot product, product rounded then sum rounded; panics on length mismatch."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    if a.shape != b.shape:
        raise ValueError("vectors must be the same length")  # clustering.go:150 panics
    s = np.float32(0)
    for p in (a * b).astype(np.float32):
        s = np.float32(s + p)
    return s


def FindClosestClusters(distanceMatrix, ctx: Optional[_lib.Context] = None) -> Tuple[int, int]:
    """clustering.go:119-133 -> (i, j) with i > j, or (-1, -1)."""
    ctx = ctx or default_context()
    D = np.asarray(distanceMatrix, np.float32)
    if D.ndim != 2 or D.shape[0] == 0:
        return -1, -1
    return ctx.find_closest(D)


def CalculateOptimalClusters(totalItems: int, minSize: int, maxSize: int):
    """clustering.go:168-186 -> (nClusters, err) with err None on success."""
    k, rc = _lib.calc_optimal_clusters(totalItems, minSize, maxSize)
    if rc is not None:
        if totalItems < minSize:
            return 0, "total items (%d) less than minimum cluster size (%d)" % (totalItems, minSize)
        return 0, ("cannot satisfy cluster size constraints with total items (%d), minSize (%d), and maxSize (%d)"
                   % (totalItems, minSize, maxSize))
    return k, None


def PerformClusteringWithConstraints(embeddings, productReferenceIDs: List[str], minSize: int, maxSize: int,
                                     ctx: Optional[_lib.Context] = None,
                                     update: int = _lib.UPDATE_EXACT) -> Tuple[Optional[Dict[int, List[str]]], bool]:
    """clustering.go:198-284 -> (map cluster id -> member ids, ok).  (None, False) on impossible constraints."""
    ctx = ctx or default_context()
    E = np.asarray(embeddings, np.float32)
    if E.ndim != 2:
        E = E.reshape(len(embeddings), -1)
    try:
        cid, rank, _ = ctx.cluster(E, minSize, maxSize, update)
    except _lib.ICLError as e:
        if e.code == _lib.ICL_ERR_CONSTRAINT:
            return None, False
        raise
    return clusters_as_map(cid, rank, productReferenceIDs), True


def clusters_as_map(cluster_id, member_rank, ids) -> Dict[int, List[str]]:
    """Canonical (cluster_id, member_rank) -> map[int][]string of clustering.go:265-280."""
    out: Dict[int, List[str]] = {}
    order = np.lexsort((member_rank, cluster_id))
    for i in order:
        c = int(cluster_id[i])
        if c >= 0:
            out.setdefault(c, []).append(ids[i])
    return out



# ---- dendrogram export -----------------------------------------------------------------------
def LastDendrogram(n: int, ctx: Optional[_lib.Context] = None) -> np.ndarray:
    """The merge tree of the last PerformClusteringWithConstraints / cluster call as a scipy-style linkage matrix
    Z[t] = [id_a, id_b, ward_distance, size]: singleton i has id i, the cluster created by merge t has id n+t (the
    engine's creation ids are exactly that convention).  The reference stops at k clusters (clustering.go:220), so Z has
    n-k rows: a forest cut at the size constraints, not a full tree.  Heights are the values FindClosestClusters
    returned (clustering.go:123-131), i.e. WardDistance of the merged pair (:84) -- not scipy's sqrt(2*d) scale."""
    ctx = ctx or default_context()
    m = ctx.last_merges().astype(np.int64)
    v = ctx.last_merge_values().astype(np.float64)
    size = np.ones(n + len(m), np.int64)
    Z = np.zeros((len(m), 4), np.float64)
    for t, (a, b) in enumerate(m):
        size[n + t] = size[a] + size[b]
        Z[t] = (a, b, v[t], size[n + t])
    return Z


def ExportDendrogram(path: str, Z: np.ndarray, ids: Optional[List[str]] = None):
    """Writes the linkage matrix (and the item ids) as JSON: {"n", "ids", "merges": [[a, b, height, size], ...]}."""
    import json

    n = int(len(ids)) if ids is not None else int(Z[:, :2].max() + 2 - len(Z)) if len(Z) else 0
    with open(path, "w") as f:
        json.dump({"n": n, "ids": list(ids) if ids is not None else None,
                   "merges": [[int(r[0]), int(r[1]), float(r[2]), int(r[3])] for r in Z]}, f)

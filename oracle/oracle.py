"""ctypes loader for the CPU oracle (oracle/libicl_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under imageclust_amd/ imports this module.  See oracle/ward_ref.c and oracle/resnet_ref.c for the
reference file:line each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libicl_oracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(_HERE, f) for f in ("ward_ref.c", "ward_fast.c", "resnet_ref.c")]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs if os.path.exists(s)):
        build()
    # team size of the OpenMP regions (ward_fast.c, resnet_ref.c): this job's CPU share, not the host's thread count
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    L = C.CDLL(_SO)
    L.icl_fast_set_threads.restype = None
    L.icl_fast_set_threads.argtypes = [C.c_int]
    L.icl_fast_set_threads(int(os.environ.get("OMP_NUM_THREADS", cores)))
    L.icl_ref_dot.restype = C.c_float
    L.icl_ref_dot.argtypes = [f32p, f32p, C.c_int64]
    L.icl_ref_ward_distance.restype = C.c_float
    L.icl_ref_ward_distance.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.c_int64]
    L.icl_ref_merge_centroid.restype = None
    L.icl_ref_merge_centroid.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.c_int64, f32p]
    L.icl_ref_initial_distance_matrix.restype = None
    L.icl_ref_initial_distance_matrix.argtypes = [f32p, C.c_void_p, C.c_int64, C.c_int64, f32p, C.c_int64]
    L.icl_ref_find_closest.restype = None
    L.icl_ref_find_closest.argtypes = [f32p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.icl_ref_calc_optimal_clusters.restype = C.c_int
    L.icl_ref_calc_optimal_clusters.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    L.icl_ref_cluster.restype = C.c_int
    L.icl_ref_cluster.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, i32p, i32p,
                                  C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.icl_fast_cluster.restype = C.c_int
    L.icl_fast_cluster.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, i32p, i32p, C.POINTER(C.c_int32),
                                   C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.icl_ref_resnet50_topology.restype = C.c_int
    L.icl_ref_resnet50_topology.argtypes = [C.c_void_p]
    L.icl_ref_resnet50_forward.restype = C.c_int
    L.icl_ref_resnet50_forward.argtypes = [C.c_void_p, C.c_int64, u8p, C.c_void_p, C.c_void_p]
    L.icl_ref_conv2d.restype = None
    L.icl_ref_conv2d.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.c_int, f32p, C.c_int, C.c_int]
    L.icl_ref_bn_act.restype = None
    L.icl_ref_bn_act.argtypes = [f32p, C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.c_float, C.c_void_p, C.c_int]
    L.icl_ref_maxpool3x3s2.restype = None
    L.icl_ref_maxpool3x3s2.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int]
    L.icl_ref_global_avgpool.restype = None
    L.icl_ref_global_avgpool.argtypes = [f32p, C.c_int, C.c_int, f32p]
    L.icl_ref_fc.restype = None
    L.icl_ref_fc.argtypes = [f32p, C.c_int, f32p, f32p, C.c_int, f32p]
    L.icl_ref_preprocess_rgb_u8.restype = None
    L.icl_ref_preprocess_rgb_u8.argtypes = [u8p, f32p]
    _lib = L
    return L


# ---- clustering.go ---------------------------------------------------------------------------------
def ward_distance(ca, sa, cb, sb):
    ca = np.ascontiguousarray(ca, np.float32)
    cb = np.ascontiguousarray(cb, np.float32)
    return np.float32(lib().icl_ref_ward_distance(ca, int(sa), cb, int(sb), ca.shape[0]))


def merge_centroid(ca, sa, cb, sb):
    ca = np.ascontiguousarray(ca, np.float32)
    cb = np.ascontiguousarray(cb, np.float32)
    out = np.empty_like(ca)
    lib().icl_ref_merge_centroid(ca, int(sa), cb, int(sb), ca.shape[0], out)
    return out


def initial_distance_matrix(centroids, sizes=None):
    Cm = np.ascontiguousarray(centroids, np.float32)
    n, d = Cm.shape
    D = np.zeros((n, n), np.float32)
    sp = None
    if sizes is not None:
        sizes = np.ascontiguousarray(sizes, np.int32)
        sp = sizes.ctypes.data
    lib().icl_ref_initial_distance_matrix(Cm, sp, n, d, D, n)
    return D


def find_closest(D):
    D = np.ascontiguousarray(D, np.float32)
    n = D.shape[0]
    i, j = C.c_int64(), C.c_int64()
    lib().icl_ref_find_closest(D.reshape(-1) if n else np.zeros(1, np.float32), n, D.shape[1] if n else 0,
                               C.byref(i), C.byref(j))
    return int(i.value), int(j.value)


def calc_optimal_clusters(total, min_size, max_size):
    k = C.c_int64()
    rc = lib().icl_ref_calc_optimal_clusters(total, min_size, max_size, C.byref(k))
    return (int(k.value), None) if rc == 0 else (0, rc)


def cluster(E, min_size, max_size, want_log=False, threads=None):
    """Returns dict(ok, cluster_id, member_rank, n_clusters, merges, skips, log).  ward_ref.c is single-threaded by
    construction (no OpenMP pragma: the reference clusters on one goroutine); `threads` is accepted for symmetry."""
    E = np.ascontiguousarray(E, np.float32)
    n, d = E.shape
    cid = np.full(max(n, 1), -1, np.int32)
    rank = np.full(max(n, 1), -1, np.int32)
    nc = C.c_int32()
    nm, ns = C.c_int64(), C.c_int64()
    log = np.zeros((max(n, 1), 4), np.int64) if want_log else None
    rc = lib().icl_ref_cluster(E if n else np.zeros((1, max(d, 1)), np.float32), n, d, min_size, max_size, cid, rank,
                               C.byref(nc), log.ctypes.data if want_log else None, C.byref(nm), C.byref(ns))
    return dict(ok=(rc == 0), rc=rc, cluster_id=cid[:n], member_rank=rank[:n], n_clusters=int(nc.value),
                merges=int(nm.value), skips=int(ns.value), log=(log[: nm.value] if want_log else None))


def cluster_fast(E, min_size, max_size, lazy_ban=True, want_log=True):
    """ward_fast.c: the sub-cubic restatement (O(n*D) per merge) for sizes ward_ref.c cannot reach.  Same outputs as
    cluster() plus vals (the Ward value of every merged pair).  log rows: pos_i, pos_j, creation_id_i, creation_id_j."""
    E = np.ascontiguousarray(E, np.float32)
    n, d = E.shape
    cid = np.full(max(n, 1), -1, np.int32)
    rank = np.full(max(n, 1), -1, np.int32)
    nc = C.c_int32()
    nm, ns = C.c_int64(), C.c_int64()
    log = np.zeros((max(n, 1), 4), np.int64) if want_log else None
    vals = np.zeros(max(n, 1), np.float32)
    rc = lib().icl_fast_cluster(E if n else np.zeros((1, max(d, 1)), np.float32), n, d, min_size, max_size, 1 if lazy_ban else 0,
                                cid, rank, C.byref(nc), log.ctypes.data if want_log else None, vals.ctypes.data, C.byref(nm), C.byref(ns))
    return dict(ok=(rc == 0), rc=rc, cluster_id=cid[:n], member_rank=rank[:n], n_clusters=int(nc.value), merges=int(nm.value),
                skips=int(ns.value), log=(log[: nm.value] if want_log else None), vals=vals[: nm.value])


def clusters_as_map(cluster_id, member_rank, ids):
    """Canonical (cluster_id, member_rank) -> the reference's map[int][]string (clustering.go:265-280)."""
    out = {}
    order = np.lexsort((member_rank, cluster_id))
    for i in order:
        c = int(cluster_id[i])
        if c < 0:
            continue
        out.setdefault(c, []).append(ids[i])
    return out


# ---- embeddings.go ---------------------------------------------------------------------------------
def resnet50_forward(blob: bytes, img_hwc_rgb_u8):
    img = np.ascontiguousarray(img_hwc_rgb_u8, np.uint8).reshape(224, 224, 3)
    pooled = np.zeros(2048, np.float32)
    dense = np.zeros(1000, np.float32)
    buf = (C.c_char * len(blob)).from_buffer_copy(blob) if not isinstance(blob, np.ndarray) else None
    ptr = C.addressof(buf) if buf is not None else blob.ctypes.data
    nbytes = len(blob) if buf is not None else blob.nbytes
    rc = lib().icl_ref_resnet50_forward(ptr, nbytes, img, pooled.ctypes.data, dense.ctypes.data)
    if rc:
        raise RuntimeError("icl_ref_resnet50_forward rc=%d" % rc)
    return pooled, dense

"""Pins the CPU oracle (oracle/ward_ref.c) to the hand-derived known-answer tests of SURVEY.md 8c
(the reference, /root/reference/internal/clustering/clustering.go, ships no tests of its own), and checks it
against two independent implementations: a numpy creation-id restatement and scipy's Ward linkage."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import ward_numpy as WN

MAXF = np.finfo(np.float32).max


def ids(n):
    return ["a%d" % i for i in range(n)]


def run_map(E, mn, mx):
    E = np.asarray(E, np.float32).reshape(len(E), -1)
    r = O.cluster(E, mn, mx, want_log=True)
    return r, (O.clusters_as_map(r["cluster_id"], r["member_rank"], ids(len(E))) if r["ok"] else None)


def test_kat1_tiebreak_and_member_order():
    r, m = run_map([[0], [1], [3], [7], [8], [20]], 1, 2)
    assert r["ok"] and m == {0: ["a2"], 1: ["a5"], 2: ["a1", "a0"], 3: ["a4", "a3"]}
    assert r["log"][:, :2].tolist() == [[1, 0], [2, 1]]  # positions (i, j) of the two merges


def test_kat2_maxsize_skip():
    r, m = run_map([[0], [1], [2], [10]], 2, 2)
    assert r["ok"] and m == {0: ["a1", "a0"], 1: ["a3", "a2"]}
    assert r["skips"] == 1 and r["merges"] == 2


def test_kat3_ban_sequence():
    r, m = run_map([[0], [1], [3], [7], [8], [20]], 2, 2)
    assert r["ok"] and m == {0: ["a1", "a0"], 1: ["a4", "a3"], 2: ["a5", "a2"]}
    assert r["skips"] == 4 and r["merges"] == 3


def test_kat3b_minsize_drop():
    r, m = run_map([[0], [1], [2], [100]], 2, 3)
    assert r["ok"] and m == {0: ["a1", "a0", "a2"]}
    assert r["cluster_id"].tolist() == [0, 0, 0, -1]
    assert r["member_rank"].tolist() == [1, 0, 2, -1]


def test_kat4_constraint_errors():
    assert not O.cluster(np.zeros((2, 1), np.float32), 3, 5)["ok"]
    assert not O.cluster(np.zeros((10, 1), np.float32), 4, 4)["ok"]


def test_kat5_optimal_clusters():
    assert O.calc_optimal_clusters(64, 3, 6) == (16, None)
    assert O.calc_optimal_clusters(250000, 5, 50) == (27500, None)
    assert O.calc_optimal_clusters(7, 1, 7) == (4, None)
    assert O.calc_optimal_clusters(10000, 5, 50) == (1100, None)
    assert O.calc_optimal_clusters(100000, 5, 50) == (11000, None)
    assert O.calc_optimal_clusters(2, 3, 5)[1] == 1
    assert O.calc_optimal_clusters(10, 4, 4)[1] == 2
    assert O.calc_optimal_clusters(10, 0, 4)[1] == 3


def test_kat6_find_closest():
    D = np.full((5, 5), MAXF, np.float32)
    assert O.find_closest(D) == (-1, -1)
    assert O.find_closest(np.zeros((1, 1), np.float32)) == (-1, -1)
    D = np.full((4, 4), 5.0, np.float32)
    D[2, 1] = np.nan
    D[3, 0] = np.inf
    D[3, 2] = 1.0
    D[2, 0] = 1.0  # tie: (2,0) precedes (3,2) in row-major lower-triangle order
    assert O.find_closest(D) == (2, 0)
    D[0, 3] = -1.0  # upper triangle is never read
    assert O.find_closest(D) == (2, 0)


def test_kat7_initial_matrix_symmetry_zero_diag():
    rng = np.random.default_rng(7)
    E = rng.standard_normal((33, 17)).astype(np.float32)
    D = O.initial_distance_matrix(E)
    assert np.all(np.diag(D) == 0)
    assert np.array_equal(D, D.T)
    # singleton Ward = 0.5*||a-b||^2 evaluated in the reference's order
    for (i, j) in [(5, 2), (32, 0), (17, 16)]:
        assert D[i, j] == WN.ward(E[i], 1, E[j], 1)
    sizes = rng.integers(1, 9, 33).astype(np.int32)
    D2 = O.initial_distance_matrix(E, sizes)
    assert D2[9, 4] == WN.ward(E[9], sizes[9], E[4], sizes[4])


def test_ward_distance_is_unfused_fp32():
    # values chosen so that fma(d,d,s) != round(round(d*d)+s): catches -ffp-contract leaks
    rng = np.random.default_rng(3)
    a = rng.standard_normal(2048).astype(np.float32)
    b = rng.standard_normal(2048).astype(np.float32)
    assert O.ward_distance(a, 3, b, 5) == WN.ward(a, 3, b, 5)
    assert np.array_equal(O.merge_centroid(a, 3, b, 5),
                          ((np.float32(3) * a + np.float32(5) * b) / np.float32(8)).astype(np.float32))


def mog(n, d, seed, k=None, sigma=0.1):
    rng = np.random.default_rng(seed)
    k = k or max(1, n // 20)
    cen = rng.standard_normal((k, d)).astype(np.float32)
    lab = rng.integers(0, k, n)
    return (cen[lab] + sigma * rng.standard_normal((n, d))).astype(np.float32)


@pytest.mark.parametrize("n,d,mn,mx,seed", [(40, 8, 1, 40, 0), (64, 32, 3, 6, 20250217), (50, 4, 2, 5, 1),
                                            (30, 3, 1, 2, 2), (25, 5, 5, 5, 3), (48, 16, 1, 3, 4)])
def test_oracle_equals_creation_id_restatement(n, d, mn, mx, seed):
    E = mog(n, d, seed)
    r = O.cluster(E, mn, mx)
    w = WN.cluster_creation_id(E, mn, mx)
    assert r["ok"] and w is not None
    assert np.array_equal(r["cluster_id"], w[0]) and np.array_equal(r["member_rank"], w[1]) and r["n_clusters"] == w[2]


@pytest.mark.parametrize("seed", range(4))
def test_oracle_equals_creation_id_on_exact_ties(seed):
    rng = np.random.default_rng(seed)
    E = rng.integers(0, 4, (48, 4)).astype(np.float32)  # many exactly tied distances and duplicate points
    for mn, mx in [(1, 48), (2, 6), (1, 2), (3, 4)]:
        r = O.cluster(E, mn, mx)
        w = WN.cluster_creation_id(E, mn, mx)
        assert r["ok"] == (w is not None)
        if r["ok"]:
            assert np.array_equal(r["cluster_id"], w[0]) and np.array_equal(r["member_rank"], w[1])


def test_oracle_merge_order_matches_scipy_ward():
    """Unconstrained, tie-free data: scipy linkage('ward') height = sqrt(2*ward); same merge sequence."""
    from scipy.cluster.hierarchy import linkage

    E = mog(60, 6, 11, k=6, sigma=0.5)
    n = len(E)
    # min=1,max=n -> k=(1+n)//2: compare the first n-k merges
    r = O.cluster(E, 1, n, want_log=True)
    Z = linkage(E.astype(np.float64), method="ward")
    for t in range(r["merges"]):
        a, b = sorted(int(x) for x in r["log"][t, 2:4])
        za, zb = sorted(int(x) for x in Z[t, :2])
        assert (a, b) == (za, zb), t


def test_size_bounds_and_dense_ids():
    E = mog(120, 8, 5)
    r = O.cluster(E, 3, 7)
    cid = r["cluster_id"]
    kept = cid[cid >= 0]
    assert sorted(set(kept.tolist())) == list(range(r["n_clusters"]))
    counts = np.bincount(kept)
    assert counts.min() >= 3 and counts.max() <= 7

"""Pins oracle/ward_fast.c (the sub-cubic restatement that checks the HIP engine at N > 40 000) to the literal
restatement oracle/ward_ref.c of /root/reference/internal/clustering/clustering.go: cluster ids, member ranks, the merge
log WITH compacted positions, the number of MaxFloat32 bans (:228-234) and the Ward value of every merged pair must be
identical, bit for bit, on every small input of the suite; the static size mask must equal the literal lazy ban."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import ward_cases as WC
from tests import ward_numpy as WN

CASES = WC.small_cases()


def same(r, f, skips=True):
    assert r["ok"] == f["ok"]
    assert np.array_equal(r["cluster_id"], f["cluster_id"]) and np.array_equal(r["member_rank"], f["member_rank"])
    assert r["n_clusters"] == f["n_clusters"] and r["merges"] == f["merges"]
    assert np.array_equal(r["log"], f["log"]), "merge log (positions and creation ids)"
    if skips:
        assert r["skips"] == f["skips"], "number of oversize bans"


@pytest.mark.parametrize("name,E,mn,mx", CASES, ids=[c[0] for c in CASES])
def test_fast_equals_literal(name, E, mn, mx):
    r = O.cluster(E, mn, mx, want_log=True)
    f = O.cluster_fast(E, mn, mx, lazy_ban=True)
    same(r, f)
    g = O.cluster_fast(E, mn, mx, lazy_ban=False)  # static mask: no skip iterations, same everything else
    same(r, g, skips=False)
    assert g["skips"] == 0
    assert np.array_equal(f["vals"].view(np.uint32), g["vals"].view(np.uint32))


def test_fast_merge_values_are_the_ward_distances():
    E = WC.mog(180, 24, 77, k=12)
    f = O.cluster_fast(E, 2, 9)
    cen = {i: E[i].copy() for i in range(len(E))}
    size = {i: 1 for i in range(len(E))}
    for t, (a, b) in enumerate(f["log"][:, 2:4].tolist()):
        assert np.float32(f["vals"][t]).view(np.uint32) == O.ward_distance(cen[a], size[a], cen[b], size[b]).view(np.uint32)
        cen[len(E) + t] = O.merge_centroid(cen[a], size[a], cen[b], size[b])
        size[len(E) + t] = size[a] + size[b]


def test_fast_constraint_errors():
    assert O.cluster_fast(np.zeros((2, 1), np.float32), 3, 5)["rc"] == 1
    assert O.cluster_fast(np.zeros((10, 1), np.float32), 4, 4)["rc"] == 2
    assert O.cluster_fast(np.zeros((10, 1), np.float32), 0, 4)["rc"] == 3
    f = O.cluster_fast(np.zeros((0, 4), np.float32), 1, 1)
    assert f["rc"] == 1 or f["n_clusters"] == 0


def test_fast_equals_literal_mid_size():
    """N=1200 D=64 min=5 max=50 (the largest input the literal oracle checks the engine on): ~2 s of O(N^3) scan."""
    E = WC.mog(1200, 64, 42)
    same(O.cluster(E, 5, 50, want_log=True), O.cluster_fast(E, 5, 50))


def test_lazy_ban_equals_static_mask_beyond_the_literal_oracle():
    """N=4000 with tight size limits (up to 3*10^5 bans): the literal ban and the static mask must agree -- this is the
    claim 'the MaxFloat32 ban is a memo of size_p + size_q > maxSize' checked where ward_ref.c is too slow."""
    for E, mn, mx in [(WC.mog(4000, 8, 5), 5, 6), (WC.mog(3000, 8, 5, sigma=1.0), 4, 4), (WC.ties(3000, 3, 9, levels=6), 6, 7)]:
        f = O.cluster_fast(E, mn, mx, lazy_ban=True)
        g = O.cluster_fast(E, mn, mx, lazy_ban=False)
        assert f["skips"] > 50
        same(f, g, skips=False)
        assert np.array_equal(f["vals"].view(np.uint32), g["vals"].view(np.uint32))


def test_fast_equals_numpy_creation_id_restatement():
    E = WC.ties(300, 3, 21, levels=5)
    f = O.cluster_fast(E, 2, 7)
    w = WN.cluster_creation_id(E, 2, 7)
    assert np.array_equal(f["cluster_id"], w[0]) and np.array_equal(f["member_rank"], w[1]) and f["n_clusters"] == w[2]

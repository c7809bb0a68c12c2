"""FAST mode (north_star's literal path): MFMA distance tile (distance_mfma.hip) + Lance-Williams updates.
It is NOT bit-identical to clustering.go by construction (GEMM-form distances, LW instead of the centroid recompute),
so these tests bound the distance error against fp64 and REPORT id agreement with the exact mode instead of asserting
equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from imageclust_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


def mog(n, d, seed, k=None, sigma=0.1):
    rng = np.random.default_rng(seed)
    k = k or max(1, n // 20)
    cen = rng.standard_normal((k, d)).astype(np.float32)
    lab = rng.integers(0, k, n)
    return (cen[lab] + sigma * rng.standard_normal((n, d))).astype(np.float32), lab


@pytest.mark.parametrize("n,d", [(1, 64), (130, 64), (257, 100), (300, 2048), (129, 1000)])
def test_mfma_distance_tile_error_bound(ctx, n, d):
    E, _ = mog(n, d, n + d)
    D = ctx.distance_mfma(E)
    E64 = E.astype(np.float64)
    sq = (E64 * E64).sum(1)
    ref = np.tril(0.5 * (sq[:, None] + sq[None, :]) - E64 @ E64.T, -1)
    scale = 0.5 * (sq[:, None] + sq[None, :])
    err = np.abs(np.tril(D, -1) - ref) / np.maximum(scale, 1e-30)
    print("n=%d d=%d max |err| / (0.5(|a|^2+|b|^2)) = %.3g" % (n, d, err.max() if n > 1 else 0.0))
    assert np.all(np.diag(D) == 0)
    if n > 1:
        assert err.max() < 2e-5  # bf16x3 split operands, fp32 accumulate


def rand_index(a, b):
    from itertools import combinations

    same = 0
    tot = 0
    idx = np.random.default_rng(0).choice(len(a), min(len(a), 400), replace=False)
    for i, j in combinations(idx, 2):
        tot += 1
        same += ((a[i] == a[j] and a[i] >= 0) == (b[i] == b[j] and b[i] >= 0))
    return same / max(tot, 1)


@pytest.mark.parametrize("n,d,mn,mx", [(300, 64, 3, 30), (1000, 256, 5, 50), (2000, 2048, 5, 50)])
def test_lw_mode_agrees_with_exact_on_separated_data(ctx, n, d, mn, mx):
    from imageclust_amd import _lib

    E, lab = mog(n, d, n, k=n // 20)
    cid_e, rank_e, nc_e = ctx.cluster(E, mn, mx, _lib.UPDATE_EXACT)
    cid_f, rank_f, nc_f = ctx.cluster(E, mn, mx, _lib.UPDATE_LW)
    kept = cid_f[cid_f >= 0]
    counts = np.bincount(kept)
    assert counts.min() >= mn and counts.max() <= mx
    assert sorted(set(kept.tolist())) == list(range(nc_f))
    ri = rand_index(cid_e, cid_f)
    print("n=%d: exact %d clusters, fast %d clusters, identical ids: %.4f, Rand index %.4f" % (n, nc_e, nc_f, float((cid_e == cid_f).mean()), ri))
    assert ri > 0.98 and abs(nc_e - nc_f) <= max(2, nc_e // 20)


def test_lw_mode_constraint_errors(ctx):
    from imageclust_amd import _lib

    with pytest.raises(_lib.ICLError) as ei:
        ctx.cluster(np.zeros((10, 4), np.float32), 4, 4, _lib.UPDATE_LW)
    assert ei.value.code == _lib.ICL_ERR_CONSTRAINT
    with pytest.raises(_lib.ICLError):
        ctx.cluster(np.zeros((10, 4), np.float32), 1, 4, 7)

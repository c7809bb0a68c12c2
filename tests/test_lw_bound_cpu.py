"""The Lance-Williams lower bound behind ICL_DIST_LWBOUND (imageclust_amd/csrc/ward.hip: ward_lb_value, ward_lb_consts_kernel), restated in
numpy float32 with the kernel's operation order and constants, against the oracle's WardDistance (oracle/ward_ref.c, clustering.go:136-157)
on the fp32 centroids the oracle's MergeClusters (clustering.go:37-40) produces.  The engine never lets a bound reach a comparison, so a
wrong bound would only show as a rare wrong merge; this test checks the inequality L(c, x) <= R(c, x) itself -- over many generations of
bounds built from bounds, near-duplicate points, large common offsets (big norms, small distances) and mixed sizes.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O

U = np.float64(2.0) ** -24


def consts(E, d, max_size):
    """ward_lb_consts_kernel: g1 and 2.01 * Delta from the centred norms and the mean (here: computed in float64, rounded up like the kernel)"""
    n = len(E)
    mu = E.astype(np.float64).mean(axis=0)
    nrm = ((E.astype(np.float64) - mu) ** 2).sum(axis=1)
    depth = min(max_size, n)
    M = (np.sqrt(nrm.max()) + np.sqrt((mu * mu).sum())) * (1.0 + 6.0 * U * depth) * 1.001
    g = np.expm1((d + 8) * np.log1p(U))
    g1 = np.float32((1.01 * g + 16.0 * U) * (1.0 + 1e-6))
    delta2 = np.float32(2.01 * (4.01 * np.sqrt(2.0) * U * M) * (1.0 + 1e-6))
    return g1, delta2


def lb_value(La, Lb, Rab, sa, sb, sx, g1, delta2):
    """ward_lb_value, vectorised over x (float32 throughout, the kernel's order of operations)"""
    f = np.float32
    La, Lb, sx = La.astype(f), Lb.astype(f), sx.astype(np.int64)
    st = (sa + sb + sx).astype(f)
    p = (sa + sx).astype(f) * La + (sb + sx).astype(f) * Lb
    q = sx.astype(f) * f(Rab)
    wl = (p * (f(1.0) - g1) - q * (f(1.0) + g1)) / st
    w = ((sa + sb) * sx).astype(f) / st
    with np.errstate(invalid="ignore"):
        L = wl * (f(1.0) - g1) - delta2 * np.sqrt(w * np.maximum(wl, f(0.0)))
    ok = (wl > 0) & (L > f(1e-30)) & (L < f(1e37))
    return np.where(ok, L, f(0.0)).astype(f)


def run_generations(E, max_size, rng, merges, start_slack=0.0):
    """Random merges of live clusters; the bound matrix is never refreshed: bounds are built from bounds for `merges` generations."""
    n, d = E.shape
    g1, delta2 = consts(E, d, max_size)
    cent = [E[i].copy() for i in range(n)]
    size = [1] * n
    live = list(range(n))
    R = {}  # exact values of live pairs (oracle)
    L = {}  # lower bounds
    for i in range(n):
        for j in range(i):
            R[(i, j)] = O.ward_distance(cent[i], 1, cent[j], 1)
            L[(i, j)] = np.float32(R[(i, j)] * (1.0 - start_slack))
    key = lambda p, q: (p, q) if p > q else (q, p)
    worst_gap = 0.0
    for _ in range(merges):
        if len(live) < 3:
            break
        ia, ib = rng.choice(len(live), 2, replace=False)
        a, b = live[ia], live[ib]
        if size[a] + size[b] > max_size:
            continue
        c = len(cent)
        cent.append(O.merge_centroid(cent[a], size[a], cent[b], size[b]))
        size.append(size[a] + size[b])
        others = [x for x in live if x != a and x != b]
        xs = np.array(others)
        La = np.array([L[key(a, x)] for x in others], np.float32)
        Lb = np.array([L[key(b, x)] for x in others], np.float32)
        sx = np.array([size[x] for x in others])
        Lc = lb_value(La, Lb, R[key(a, b)], size[a], size[b], sx, g1, delta2)
        for x, l in zip(others, Lc):
            r = O.ward_distance(cent[c], size[c], cent[x], size[x])
            assert l <= r, "bound %r above the reference's value %r (sizes %d+%d vs %d, generation of c: %d)" % (l, r, size[a], size[b], size[x], size[c])
            if r > 0:
                worst_gap = max(worst_gap, (r - l) / r)
            R[key(c, x)] = r
            L[key(c, x)] = l
        live = others + [c]
    return worst_gap


@pytest.mark.parametrize("seed,n,d", [(0, 40, 32), (1, 30, 256), (2, 24, 2048), (3, 48, 8)])
def test_bound_holds_over_generations_on_generic_points(seed, n, d):
    rng = np.random.default_rng(seed)
    E = rng.standard_normal((n, d)).astype(np.float32)
    gap = run_generations(E, 1000, rng, 3 * n)
    assert gap < 0.2  # ~3 g per generation: loose would mean a useless bound, not a wrong one


def test_bound_holds_with_a_large_common_offset_and_tiny_distances():
    """ResNet-like: non-negative features with a large mean, points that differ in the last bits -- the centroid-rounding term (Delta) decides."""
    rng = np.random.default_rng(7)
    base = (100.0 + rng.standard_normal((1, 64))).astype(np.float32)
    E = (base + 1e-4 * rng.standard_normal((36, 64))).astype(np.float32)
    run_generations(E, 1000, rng, 100)
    E2 = (base * (1 + 1e-7 * rng.integers(-3, 4, (30, 64)))).astype(np.float32)  # differences of a few ulps: most bounds must come out 0
    run_generations(E2, 1000, rng, 80)


def test_bound_holds_on_duplicates_ties_and_when_started_from_loose_bounds():
    rng = np.random.default_rng(11)
    E = rng.integers(0, 3, (40, 8)).astype(np.float32)  # exact ties and duplicates: values 0 and small integers
    run_generations(E, 1000, rng, 120)
    E = rng.standard_normal((32, 16)).astype(np.float32)
    run_generations(E, 6, rng, 200)                      # tight size constraint: many skipped merges, small clusters
    run_generations(E, 1000, rng, 90, start_slack=0.01)  # the initial matrix's bounds sit below the values too

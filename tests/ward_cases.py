"""Seeded input generators shared by the Ward tests (CPU oracle tests and -m gpu parity tests)."""
import numpy as np


def mog(n, d, seed, k=None, sigma=0.1):
    """Mixture of Gaussians (SURVEY.md 8d 'Synthetic embeddings'): k = n/20 centres, points = centre + sigma*N(0,1)."""
    rng = np.random.default_rng(seed)
    k = k or max(1, n // 20)
    cen = rng.standard_normal((k, d)).astype(np.float32)
    lab = rng.integers(0, k, n)
    return (cen[lab] + sigma * rng.standard_normal((n, d))).astype(np.float32)


def ties(n, d, seed, levels=4):
    """Small-integer coordinates: many exactly tied distances and duplicate points."""
    return np.random.default_rng(seed).integers(0, levels, (n, d)).astype(np.float32)


def hub_and_spokes(d=48):
    """A hub that absorbs one spoke after another: almost every merge involves the cluster the previous one created."""
    E = np.zeros((d + 1, d), np.float32)
    for i in range(d):
        E[i + 1, i] = 1.1 ** i
    return E


def quadruples(seed=11, groups=60, d=12):
    """Tight pairs of pairs: clusters created back to back are each other's nearest neighbours."""
    rng = np.random.default_rng(seed)
    base = (rng.standard_normal((groups, d)) * 50).astype(np.float32)
    off = np.zeros((4, d), np.float32)
    off[1, 0] = 0.01
    off[2, 0] = 0.05
    off[3, 0] = 0.061
    E = (base[:, None, :] + off[None, :, :]).reshape(-1, d)
    return E[rng.permutation(len(E))]


def small_cases():
    """(name, E, min, max): every small clustering input of the suite, for oracle-vs-oracle and engine-vs-oracle checks."""
    out = [("kat1", np.array([[0], [1], [3], [7], [8], [20]], np.float32), 1, 2),
           ("kat2", np.array([[0], [1], [2], [10]], np.float32), 2, 2),
           ("kat3", np.array([[0], [1], [3], [7], [8], [20]], np.float32), 2, 2),
           ("kat3b", np.array([[0], [1], [2], [100]], np.float32), 2, 3)]
    for (n, d, mn, mx, seed) in [(40, 8, 1, 40, 0), (64, 32, 3, 6, 20250217), (50, 4, 2, 5, 1), (30, 3, 1, 2, 2), (25, 5, 5, 5, 3),
                                 (48, 16, 1, 3, 4), (200, 64, 3, 6, 5), (2, 4, 1, 2, 8), (1, 4, 1, 1, 9), (65, 7, 1, 1, 10),
                                 (130, 100, 1, 130, 7), (333, 20, 5, 50, 6)]:
        out.append(("mog_%d_%d_%d_%d" % (n, d, mn, mx), mog(n, d, seed), mn, mx))
    for seed in range(4):
        for mn, mx in [(1, 48), (2, 6), (1, 2), (3, 4)]:
            out.append(("ties_s%d_%d_%d" % (seed, mn, mx), ties(48, 4, seed), mn, mx))
    for mn, mx in [(1, 49), (1, 12), (2, 5)]:
        out.append(("hub_%d_%d" % (mn, mx), hub_and_spokes(), mn, mx))
    x = np.cumsum(1.07 ** np.arange(150)).astype(np.float32)[:, None]
    for mn, mx in [(1, 150), (2, 9), (1, 3)]:
        out.append(("gaps_%d_%d" % (mn, mx), x, mn, mx))
    for mn, mx in [(1, 240), (2, 4), (3, 8), (1, 2)]:
        out.append(("quads_%d_%d" % (mn, mx), quadruples(), mn, mx))
    for mn, mx in [(1, 600), (2, 12)]:
        out.append(("ties600_%d_%d" % (mn, mx), ties(600, 5, 3, levels=3), mn, mx))
    out.append(("identical", np.ones((37, 5), np.float32), 2, 4))
    E = mog(40, 6, 3)
    E[7, 2] = np.nan
    E[11, 0] = np.inf
    out.append(("nan_inf", E, 1, 3))
    out.append(("no_pairs_left", mog(20, 3, 1), 1, 1))
    for n, mn, mx in [(100, 1, 1), (100, 1, 2), (101, 1, 2), (37, 1, 3), (64, 5, 64)]:
        out.append(("target_%d_%d_%d" % (n, mn, mx), mog(n, 16, n), mn, mx))
    return out


def random_case(rng):
    """One input of the randomised sweeps (scratch/lb_sweep.py, test_randomised_sweep_*): size, dimension (multiples of 4: the Lance-Williams bound
    rows need whole k-groups), data shape and constraints drawn at random.  Shapes: mixture of Gaussians; non-negative features with a large common
    mean and a small spread (ResNet-like, and worse); integer grids (exact ties, duplicates); rank 3; heavy tails."""
    n = int(rng.choice([300, 900, 1500, 2500, 4200, 6000]))
    d = int(rng.choice([4, 8, 16, 64, 128, 512, 2048]))
    if d == 2048 and n > 2500:
        n = 2500
    kind = str(rng.choice(["mog", "offset", "grid", "lowrank", "heavy"]))
    if kind == "mog":
        k = max(n // int(rng.choice([5, 20, 60])), 1)
        cen = rng.standard_normal((k, d)).astype(np.float32)
        E = cen[rng.integers(0, k, n)] + np.float32(rng.choice([0.02, 0.1, 0.5])) * rng.standard_normal((n, d)).astype(np.float32)
    elif kind == "offset":
        E = (np.float32(rng.choice([5.0, 50.0])) + np.abs(rng.standard_normal((1, d))).astype(np.float32)
             + np.float32(rng.choice([1e-3, 0.05])) * rng.standard_normal((n, d)).astype(np.float32))
    elif kind == "grid":
        E = rng.integers(0, int(rng.choice([2, 3, 6])), (n, d)).astype(np.float32)
    elif kind == "lowrank":
        E = (rng.standard_normal((n, 3)) @ rng.standard_normal((3, d))).astype(np.float32)
    else:
        E = (rng.standard_normal((n, d)) * np.exp(2.0 * rng.standard_normal((n, 1)))).astype(np.float32)
    mn = int(rng.choice([1, 2, 5]))
    mx = max(int(rng.choice([mn, mn + 1, 6, 50, 1000])), mn)
    return kind, np.ascontiguousarray(E, np.float32), mn, mx

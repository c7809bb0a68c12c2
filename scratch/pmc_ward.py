"""Ward path with eager launches (prof mask on the update kernel disables the hipGraph replay) for rocprofv3 --pmc."""
import sys
import numpy as np
sys.path.insert(0, ".")
from imageclust_amd import _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(20250217)
n, d = 10000, 2048
cen = rng.standard_normal((500, d)).astype(np.float32)
E = (cen[rng.integers(0, 500, n)] + 0.1 * rng.standard_normal((n, d))).astype(np.float32)
ctx.prof_enable(1 << _lib.K_UPDATE)
cid, rank, nc = ctx.cluster(E, 5, 50)
print("clusters", nc, "merges", len(ctx.last_merges()))

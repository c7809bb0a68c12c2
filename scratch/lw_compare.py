"""FAST mode: the batched loop must reproduce the one-merge-per-step Lance-Williams loop bit for bit (run twice:
ICL_WARD_BATCH=1 / 0, the switch is read once per process) -- writes the merge log + heights to the given file."""
import sys
import numpy as np
sys.path.insert(0, ".")
from imageclust_amd import _lib
out = sys.argv[1]
ctx = _lib.Context(0)
rng = np.random.default_rng(7)
res = {}
for n, d, mn, mx in [(3000, 64, 5, 50), (700, 16, 2, 9), (2000, 2048, 3, 30)]:
    cen = rng.standard_normal((max(n // 20, 1), d)).astype(np.float32)
    E = (cen[rng.integers(0, len(cen), n)] + 0.1 * rng.standard_normal((n, d))).astype(np.float32)
    cid, rank, nc = ctx.cluster(E, mn, mx, _lib.UPDATE_LW)
    res["m%d" % n] = ctx.last_merges()
    res["v%d" % n] = ctx.last_merge_values()
    res["c%d" % n] = cid
np.savez(out, **res)
print("saved", out)

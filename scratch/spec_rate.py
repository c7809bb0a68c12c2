"""How often is merge t+1 independent of the cluster created by merge t? (decides whether two merges per step pay)"""
import sys, numpy as np
sys.path.insert(0, ".")
from imageclust_amd import _lib
ctx = _lib.Context(0); ctx.load_synthetic(1)
n = 10000
imgs = np.concatenate([_lib.synth_images(20250217, i, 1000, _lib.SYNTH_STRUCTURED) for i in range(0, n, 1000)])
E = ctx.embed_u8(imgs, _lib.HEAD_POOLED, _lib.PREC_BF16)
ctx.cluster(E, 5, 50)
m = ctx.last_merges().astype(np.int64)
c = n + np.arange(len(m))
dep = (m[1:, 0] == c[:-1]) | (m[1:, 1] == c[:-1])
print("bench data: merges", len(m), "merge t+1 uses c_t:", dep.mean())
for q in range(0, len(dep), len(dep) // 8): print("  window", q, dep[q:q + len(dep) // 8].mean())
dep2 = dep[1:] | (m[2:, 0] == c[:-2]) | (m[2:, 1] == c[:-2])
print("merge t+2 uses c_t or c_t+1:", dep2.mean())
rng = np.random.default_rng(0)
cen = rng.standard_normal((500, 2048)).astype(np.float32)
E2 = (cen[rng.integers(0, 500, n)] + 0.1 * rng.standard_normal((n, 2048))).astype(np.float32)
ctx.cluster(E2, 5, 50)
m = ctx.last_merges().astype(np.int64); dep = (m[1:, 0] == c[:-1]) | (m[1:, 1] == c[:-1])
print("MoG data: merge t+1 uses c_t:", dep.mean())

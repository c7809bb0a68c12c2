"""How wide is the refinement band of an MFMA-bounded distance matrix?  For sampled rows of the benchmark's
embeddings (and of the mixture-of-Gaussians test input) count the entries whose LOWER bound does not exceed the row's smallest
UPPER bound, with the proven error bound of an f32-MFMA GEMM form:  |D~ - S/2| <= (12u + gamma_D / 2)(|a|^2 + |b|^2),
reference value within (1 +- gamma')(S/2), gamma_D = D u, gamma' = (D + 2) u, u = 2^-24.  Those entries would have to be
evaluated exactly (sequential fp32, clustering.go:136-157) whenever the row is (re)minimised."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from imageclust_amd import _lib  # noqa: E402
from tests.ward_pipeline_child import make_E  # noqa: E402

u = 2.0 ** -24


def probe(E, name, centre):
    n, d = E.shape
    X = E.double()
    if centre:
        X = X - X.mean(0, keepdim=True)
    nrm = (X * X).sum(1)
    g = torch.Generator(device="cpu").manual_seed(1)
    rows = torch.randint(1000, n, (1500,), generator=g).cuda()
    eps_c = 12 * u + d * u / 2
    gam = (d + 2) * u * 1.01
    cnts = []
    for r in rows.tolist():
        S2 = 0.5 * ((X[:r] - X[r]) ** 2).sum(1)  # the true S/2 against every earlier singleton
        e = eps_c * (nrm[:r] + nrm[r])
        L = torch.clamp((S2 - 2 * e) * (1 - gam), min=0)  # D~ may sit eps below the truth, the bound eps below D~
        U = (S2 + 2 * e) * (1 + gam)
        cnts.append(int((L <= U.min()).sum()))
    c = np.array(cnts)
    print("%-34s centred=%d  band entries per row scan: median %d  mean %.1f  p90 %d  max %d   (row length ~%d)" %
          (name, centre, np.median(c), c.mean(), np.percentile(c, 90), c.max(), n // 2))


ctx = _lib.Context(0)
ctx.load_synthetic(1)
n = 30000
imgs = torch.empty(n * _lib.IMG_BYTES, dtype=torch.uint8, device="cuda")
ctx.synth_images_dev(20250217, 0, n, _lib.SYNTH_STRUCTURED, imgs.data_ptr())
E = torch.empty((n, 2048), dtype=torch.float32, device="cuda")
ctx.embed_u8_dev(imgs.data_ptr(), n, E.data_ptr(), 2048, _lib.PREC_BF16)
ctx.sync()
for c in (0, 1):
    probe(E, "ResNet50 embeddings (structured)", c)
    probe(make_E(n, 2048, 20250217), "mixture of Gaussians (tests)", c)

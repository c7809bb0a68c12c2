"""Generates the committed golden fixtures from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference cannot be executed here (Go + OpenCV) and
ships no fixtures, so these vectors come from the oracle AFTER it passed the hand-derived KATs
(tests/test_oracle_ward.py) and the torch-fp64 cross-check (tests/test_oracle_resnet.py).

Fixtures (SURVEY.md 8c "Fixtures to commit"):
  ward_mog_n64_d32.npz     mixture-of-Gaussians E (N=64, D=32, seed 20250217), min=3,max=6 -> ids, ranks, merge log
  ward_ties_n48_d4.npz     integer-valued E with many exact ties (N=48, D=4) -> same, for (1,48),(2,6),(1,2),(3,4)
  resnet50_synth_seed1.npz pooled(2048)/dense0(1000) fp32 of 4 structured synthetic images under weights seed 1
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from imageclust_amd import _lib  # noqa: E402  (host-only entry points: synthetic weights and images)

HERE = os.path.dirname(os.path.abspath(__file__))


def mog(n, d, seed):
    rng = np.random.default_rng(seed)
    k = max(1, n // 20)
    cen = rng.standard_normal((k, d)).astype(np.float32)
    lab = rng.integers(0, k, n)
    return (cen[lab] + 0.1 * rng.standard_normal((n, d))).astype(np.float32)


def main():
    E = mog(64, 32, 20250217)
    r = O.cluster(E, 3, 6, want_log=True)
    np.savez_compressed(os.path.join(HERE, "ward_mog_n64_d32.npz"), E=E, min_size=3, max_size=6, cluster_id=r["cluster_id"],
                        member_rank=r["member_rank"], n_clusters=r["n_clusters"], merges=r["log"][:, 2:4].astype(np.int32))
    rng = np.random.default_rng(48)
    E = rng.integers(0, 4, (48, 4)).astype(np.float32)
    out = dict(E=E)
    for mn, mx in [(1, 48), (2, 6), (1, 2), (3, 4)]:
        r = O.cluster(E, mn, mx, want_log=True)
        out["cid_%d_%d" % (mn, mx)] = r["cluster_id"]
        out["rank_%d_%d" % (mn, mx)] = r["member_rank"]
        out["merges_%d_%d" % (mn, mx)] = r["log"][:, 2:4].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "ward_ties_n48_d4.npz"), **out)
    blob = _lib.synthetic_blob(1)
    imgs = _lib.synth_images(20250217, 0, 4, _lib.SYNTH_STRUCTURED)
    pooled, dense = zip(*[O.resnet50_forward(blob, im) for im in imgs])
    np.savez_compressed(os.path.join(HERE, "resnet50_synth_seed1.npz"), img_seed=20250217, weight_seed=1, pooled=np.stack(pooled),
                        dense=np.stack(dense), img_checksum=np.int64(imgs.astype(np.int64).sum()),
                        blob_checksum=np.float64(np.frombuffer(blob[80:].tobytes(), np.float32).astype(np.float64).sum()))


if __name__ == "__main__":
    main()

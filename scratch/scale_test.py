"""Scale check of the exact Ward engine on one GPU (not a pytest: takes tens of seconds and tens of GB).
usage: python scratch/scale_test.py N [--lib path/to/variant.so] [--lw] [--reps R]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclust_amd import _lib
import torch

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--lw", action="store_true")
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--d", type=int, default=2048)
    ap.add_argument("--dist-mode", type=int, default=0, help="icl_set_ward_options: 0 auto, 1 exact, 2 bounds in the initial matrix, 4 + Lance-Williams bound rows")
    ap.add_argument("--real", action="store_true", help="cluster the bf16 ResNet50 embeddings of the bench's structured synthetic images instead of the mixture of Gaussians")
    a = ap.parse_args()
    if a.lib:
        _lib.SO_PATH = a.lib
    n, d = a.n, a.d
    ctx = _lib.Context(0)
    ctx.set_ward_options(a.dist_mode)
    if a.real:
        ctx.load_synthetic(1)
        imgs = torch.empty(n * _lib.IMG_BYTES, dtype=torch.uint8, device="cuda")
        ctx.synth_images_dev(20250217, 0, n, _lib.SYNTH_STRUCTURED, imgs.data_ptr())
        E = torch.empty((n, 2048), dtype=torch.float32, device="cuda")
        ctx.embed_u8_dev(imgs.data_ptr(), n, E.data_ptr(), 2048, _lib.PREC_BF16)
        del imgs
        d = 2048
    else:
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        k = n // 20
        cen = torch.randn((k, d), generator=g, device="cuda")
        lab = torch.randint(0, k, (n,), generator=g, device="cuda")
        E = (cen[lab] + 0.1 * torch.randn((n, d), generator=g, device="cuda")).contiguous()
    torch.cuda.synchronize()
    modes = [(_lib.UPDATE_EXACT, "exact")] + ([(_lib.UPDATE_LW, "lw")] if a.lw else [])
    for upd, name in modes:
        for rep in range(a.reps):
            t0 = time.time()
            cid, rank, nc = ctx.cluster_dev(E.data_ptr(), n, d, 5, 50, upd)
            dt = time.time() - t0
            st = ctx.last_stage_ms()
            kept = cid[cid >= 0]; counts = np.bincount(kept)
            m = ctx.last_merges()
            print(name, "lib", a.lib, "n", n, "time %.2fs" % dt, st, "clusters", nc, "sizes", counts.min(), counts.max(), "merges", len(m), "stats(merges,steps,single,sum_live)", ctx.last_ward_stats(),
                  "dropped", int((cid < 0).sum()), "log-crc", int(np.bitwise_xor.reduce(m.astype(np.int64).ravel() * np.arange(1, m.size + 1))), flush=True)

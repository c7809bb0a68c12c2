"""Scale check of the exact Ward engine on one GPU (not a pytest: takes tens of seconds and tens of GB)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from imageclust_amd import _lib
import torch

n = int(sys.argv[1]); d = 2048
ctx = _lib.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
k = n // 20
cen = torch.randn((k, d), generator=g, device="cuda")
lab = torch.randint(0, k, (n,), generator=g, device="cuda")
E = (cen[lab] + 0.1 * torch.randn((n, d), generator=g, device="cuda")).contiguous()
torch.cuda.synchronize()
for upd, name in [(_lib.UPDATE_EXACT, "exact"), (_lib.UPDATE_LW, "lw")]:
    t0 = time.time()
    cid, rank, nc = ctx.cluster_dev(E.data_ptr(), n, d, 5, 50, upd)
    dt = time.time() - t0
    st = ctx.last_stage_ms()
    kept = cid[cid >= 0]; counts = np.bincount(kept)
    lab_h = lab.cpu().numpy()
    # purity: fraction of kept points whose cluster's majority label equals theirs
    pur = 0
    order = np.argsort(cid, kind="stable")
    print(name, "n", n, "time %.2fs" % dt, st, "clusters", nc, "sizes", counts.min(), counts.max(), "merges", len(ctx.last_merges()), "dropped", int((cid < 0).sum()), flush=True)
    if name == "exact": cid_e = cid.copy()
print("identical ids exact vs lw: %.4f" % float((cid_e == cid).mean()))

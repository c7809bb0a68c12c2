"""lb mode (ICL_DIST_LWBOUND) against the oracle on a few inputs: python scratch/lb_try.py [--lib so] [--big] [--mode 4]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclust_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--big", action="store_true")
ap.add_argument("--skip-small", action="store_true")
ap.add_argument("--mode", type=int, default=4)
a = ap.parse_args()
if a.lib:
    _lib.SO_PATH = a.lib
from oracle import oracle as O
from tests import ward_cases as WC
ctx = _lib.Context(0)
ctx.set_ward_options(a.mode)
def check(name, E, mn, mx):
    f = O.cluster_fast(E, mn, mx, lazy_ban=False)
    t0 = time.time()
    cid, rank, nc = ctx.cluster(E, mn, mx)
    dt = time.time() - t0
    m = ctx.last_merges(); v = ctx.last_merge_values()
    want = f["log"][:, 2:4].astype(np.int32)
    ok = (np.array_equal(cid, f["cluster_id"]) and np.array_equal(rank, f["member_rank"]) and nc == f["n_clusters"]
          and len(m) == f["merges"] and np.array_equal(m, want) and np.array_equal(v.view(np.uint32), f["vals"].view(np.uint32)))
    print("%-34s n=%6d d=%5d  %s  %.2fs stages %s stats %s" % (name, E.shape[0], E.shape[1], "OK" if ok else "MISMATCH", dt, ctx.last_stage_ms(), ctx.last_ward_stats()), flush=True)
    if not ok:
        nm = min(len(m), f["merges"])
        bad = [i for i in range(nm) if not (m[i] == want[i]).all() or v.view(np.uint32)[i] != f["vals"].view(np.uint32)[i]]
        print("   differing merges:", bad[:3], "of", len(m), f["merges"], "ids equal:", np.array_equal(cid, f["cluster_id"]))
        if bad:
            i = bad[0]
            print("   gpu", m[i], v[i], "oracle", want[i], f["vals"][i])
    return ok
allok = True
if not a.skip_small:
    for name, E, mn, mx in WC.small_cases():
        if E.shape[1] % 4 == 0 and E.shape[0] >= 8:
            allok &= check(name, E, mn, mx)
    allok &= check("mog 6000x16", WC.mog(6000, 16, 1), 5, 50)
    allok &= check("ties 1100", WC.ties(1100, 4, 2, levels=5), 2, 9)
    allok &= check("quadruples 250", WC.quadruples(seed=5, groups=250), 1, 1000)
    allok &= check("mog 3000x2048", WC.mog(3000, 2048, 4), 3, 6)
    allok &= check("mog 24000x16", WC.mog(24000, 16, 1), 5, 50)
if a.big:
    allok &= check("mog 20000x2048", WC.mog(20000, 2048, 7), 5, 50)
print("ALL OK" if allok else "FAILURES")

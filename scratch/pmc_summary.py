"""Per-kernel HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters are in KB).
FETCH_SIZE is doubled as MI355X_MICROARCH.md "HBM" prescribes for wide coalesced streaming reads on gfx950 (the counter
tallies 128-B requests at 64 B); WRITE_SIZE is taken as is.  usage: python scratch/pmc_summary.py <dir with pmc_fetch/ pmc_write/>"""
import csv, glob, json, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


if __name__ == "__main__":
    root = sys.argv[1]
    fe, wr = load(root + "/pmc_fetch", "FETCH_SIZE"), load(root + "/pmc_write", "WRITE_SIZE")
    out = {"note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), means over every launch of the kernel in one bench.py run "
                   "under rocprofv3 --pmc (separate passes); the same launch path as the timed region"}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, [])
        w = wr.get(k, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        out[k] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KB_mean": fm, "WRITE_SIZE_KB_mean": wm,
                  "hbm_bytes_per_launch": 2 * fm * 1024 + wm * 1024}
    print(json.dumps(out, indent=1))

"""Per-kernel summary of one forward pass (one batch) out of a rocprofv3 kernel trace: python kernel_times.py trace.csv"""
import csv, sys, re
tr = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in tr)
first = [i for i, e in enumerate(ev) if "stem" in e[2][:40]]
a, b = first[1], first[2]  # the second batch
batch = ev[a:b]
def short(n):
    n = re.sub(r"^void ", "", n)
    return n[:60]
agg = {}
for s, e, n in batch:
    k = short(n)
    agg.setdefault(k, [0.0, 0])
    agg[k][0] += (e - s) / 1e3
    agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
for k, (us, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-62s x%-3d %9.1f us  %5.1f%%" % (k, c, us, 100 * us / tot))
print("sum of kernels %.1f us; batch span %.1f us" % (tot, (batch[-1][1] - batch[0][0]) / 1e3))
print("-- in order:")
for s, e, n in batch[:16]:
    print("  %-60s %8.1f us" % (short(n), (e - s) / 1e3))

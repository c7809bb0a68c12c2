#!/bin/bash
# one rocprofv3 --kernel-trace --stats pass of bench.py (1 step): per-kernel average durations of the current build
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $out/stats.json 2> $out/stats.err
cp $out/stats/*/*_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
rm -rf $out/stats
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:14]:
    print("%-70s calls %7s  avg %10.1f us  total %9.1f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
